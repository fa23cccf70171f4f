#!/bin/bash
# round 6: k_stream_b9 with the 32 x 32 x 16 MFMA waves (even tile and column-tile counts) - ragged shapes, the 7B launches, compute side alone (abl8)
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_b9s_9.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout 120 ./b9s_probe "$@" >> $OUT 2>&1; echo "rc $?" >> $OUT; }
run 512 1024 20 -
run 512 1024 32 -
run 704 1024 64 - 2 1
run 2048 512 50 - 2
run 2048 512 64 - 3
run 4096 512 17 - 1 0 2
run 8192 256 30 - 1 0 4
for n in 32 64; do
  run 11008 4096 $n - 2 1
  run 4096 4096 $n - 3
  run 4096 4096 $n - 1 0 4
  run 4096 11008 $n - 1 0 4
done
run 32000 4096 32 -
export B9S_SKIP_CHECK=1 B9S_NO_DMA=1
for shape in "11008 4096 64 - 2 1" "4096 11008 64 - 1 0 4" "4096 4096 64 - 1 0 4"; do
  echo "==== $shape" >> $OUT
  timeout 120 ./b9s_probe_abl8 $shape 2>&1 | grep -v "^M \|split3" >> $OUT
done
grep -v "^rc 0" $OUT
