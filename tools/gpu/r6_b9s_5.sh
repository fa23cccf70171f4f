#!/bin/bash
# round 6: k_stream_b9 v4 (wave-specialised: four loader waves, four MFMA waves with half the tiles each) on the 7B launches, with ablations
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_b9s_5.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout 120 ./b9s_probe "$@" >> $OUT 2>&1; echo "rc $?" >> $OUT; }
run 352 1024 13 -
run 352 1024 33 -
run 352 1024 40 -
run 352 1024 64 - 2 1
run 2048 512 50 - 3
run 4096 512 17 - 1 0 2
for n in 16 32 48 64; do
  run 11008 4096 $n - 2 1
  run 4096 4096 $n - 3
  run 4096 4096 $n - 1 0 4
  run 4096 11008 $n - 1 0 4
  run 4096 4096 $n - 1 0 2
  run 4096 11008 $n - 1 0 2
done
export B9S_SKIP_CHECK=1 B9S_NO_DMA=1
for shape in "11008 4096 64 - 2 1" "4096 4096 64 - 3" "11008 4096 32 - 2 1" "4096 11008 64 - 1 0 4"; do
  echo "==== $shape" >> $OUT
  for v in _abl2 _abl4 _abl8 _np6 _nogroups; do
    timeout 120 ./b9s_probe$v $shape 2>&1 | grep -v "^M \|split3" >> $OUT
  done
  B9S_IMAGES=2 timeout 120 ./b9s_probe $shape 2>&1 | grep -v "^M \|split3" >> $OUT
done
grep -v "^M \|^rc 0" $OUT
