#!/bin/bash
# same-box A/B, resident decode fp32 + block-int8, alternating processes: lib (new) against lib_old
out=${1:-gpurun_out/r6ab3}; mkdir -p $out
swap() { (cd llama.go_amd && mv lib lib_tmp && mv lib_old lib && mv lib_tmp lib_old); }
for rep in 1 2 3 4 5 6; do
  for which in new old; do
    echo -n "$which " >> $out/ab.txt
    python tools/decode_quick.py --int8 --steps 100 --reps 8 >> $out/ab.txt 2>&1
    echo -n "$which " >> $out/ab.txt
    python tools/decode_quick.py --steps 100 --reps 5 >> $out/ab.txt 2>&1
    swap
  done
done
cat $out/ab.txt
