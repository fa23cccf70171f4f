#!/bin/bash
# same-box A/B of two builds of the product libraries: llama.go_amd/lib (new) against llama.go_amd/lib_old (baseline), int8 + fp32 resident decode
# and int8 pods; interleaved runs.  usage: bash tools/gpu/r6_ab_libs.sh <outdir>
out=${1:-gpurun_out/r6ab}; mkdir -p $out
cd llama.go_amd
swap() { mv lib lib_tmp && mv lib_old lib && mv lib_tmp lib_old; }
cd ..
for rep in 1 2; do
  for which in new old; do
    echo "== $which rep $rep" >> $out/ab.txt
    python tools/decode_quick.py --int8 --steps 64 --reps 5 >> $out/ab.txt 2>&1
    python tools/decode_quick.py --steps 64 --reps 3 >> $out/ab.txt 2>&1
    python tools/bench_pods.py --int8 --pods 2,4 --steps 48 >> $out/ab.txt 2>&1
    (cd llama.go_amd && swap)
  done
done
cat $out/ab.txt
