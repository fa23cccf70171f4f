#!/bin/bash
# round 6: k_stream_b9 standalone against k_stream_dma on the 7B launches (tools/b9s_probe)
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_b9s_1.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout 120 ./b9s_probe "$@" >> $OUT 2>&1; echo "rc $?" >> $OUT; }
# correctness first: small ragged shapes
run 352 1024 13 64
run 352 1024 33 64
run 352 1024 64 64 2 1
run 352 1024 100 64
run 352 1024 128 64
run 352 1024 30 128
for n in 16 32 48 64 96 128; do
  run 11008 4096 $n 64 2 1
  run 4096 4096 $n 64 3
  run 4096 4096 $n 64 1 0 4
  run 4096 11008 $n 64 1 0 4
done
for n in 16 32; do
  run 11008 4096 $n 128 2 1
  run 4096 4096 $n 128 3
  run 4096 4096 $n 128 1 0 4
  run 4096 11008 $n 128 1 0 4
done
cat $OUT | grep -v "^   wave" | tail -150
