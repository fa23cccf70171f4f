#!/bin/bash
# round 6: k_stream_b9 (two rings, weights one chunk ahead, column parts) standalone against k_stream_dma on the 7B launches (tools/b9s_probe)
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_b9s_2.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout 120 ./b9s_probe "$@" >> $OUT 2>&1; echo "rc $?" >> $OUT; }
# correctness first: small ragged shapes
run 352 1024 13 2
run 352 1024 33 2
run 352 1024 64 2 2 1
run 352 1024 100 2
run 352 1024 128 3
run 352 1024 30 3
run 2048 512 90 2 3
for n in 16 32 48 64 96 128; do
  run 11008 4096 $n 2 2 1
  run 4096 4096 $n 2 3
  run 4096 4096 $n 2 1 0 4
  run 4096 11008 $n 2 1 0 4
done
for n in 32 64 128; do
  run 11008 4096 $n 3 2 1
  run 4096 4096 $n 3 3
  run 4096 4096 $n 3 1 0 4
  run 4096 11008 $n 3 1 0 4
done
grep -v "^   wave\|^M \|^rc 0" $OUT | tail -150
