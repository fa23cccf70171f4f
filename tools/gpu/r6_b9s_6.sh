#!/bin/bash
# round 6: k_stream_b9 with the column-halves mode (odd tile counts): wq|wk|wv of 7B and small odd shapes, standalone
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_b9s_6.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout 120 ./b9s_probe "$@" >> $OUT 2>&1; echo "rc $?" >> $OUT; }
run 352 1024 30 -
run 352 1024 64 -
run 2048 512 50 - 3
run 2048 512 64 - 3
for n in 32 64; do
  run 4096 4096 $n - 3
done
grep -v "^M \|^rc 0" $OUT
