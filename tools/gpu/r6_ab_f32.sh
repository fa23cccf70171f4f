#!/bin/bash
# same-box A/B, fp32 resident decode only, many alternations: lib (new) against lib_old
out=${1:-gpurun_out/r6abf}; mkdir -p $out
swap() { (cd llama.go_amd && mv lib lib_tmp && mv lib_old lib && mv lib_tmp lib_old); }
for rep in 1 2 3 4 5 6 7 8; do
  for which in new old; do
    echo -n "$which " >> $out/ab.txt
    python tools/decode_quick.py --steps 100 --reps 6 >> $out/ab.txt 2>&1
    swap
  done
done
cat $out/ab.txt
