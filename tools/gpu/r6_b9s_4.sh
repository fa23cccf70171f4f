#!/bin/bash
# round 6: k_stream_b9 v3 (software-pipelined operands) - 7B launches, NX = 3 (planes one chunk ahead) and NX = 2, with ablations
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_b9s_4.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout 120 ./b9s_probe "$@" >> $OUT 2>&1; echo "rc $?" >> $OUT; }
run 352 1024 13 2
run 352 1024 33 3
run 352 1024 40 2
run 352 1024 64 3 2 1
run 352 1024 100 2
run 352 1024 90 3
run 352 1024 128 2
run 2048 512 90 2 3
for n in 16 32 48 64 96 128; do
 for nx in 3 2; do
  run 11008 4096 $n $nx 2 1
  run 4096 4096 $n $nx 3
  run 4096 4096 $n $nx 1 0 4
  run 4096 11008 $n $nx 1 0 4
 done
done
export B9S_SKIP_CHECK=1 B9S_NO_DMA=1
for shape in "11008 4096 64 3 2 1" "4096 4096 64 3 3" "11008 4096 32 3 2 1"; do
  echo "==== $shape" >> $OUT
  for v in _heavy _abl4 _abl9 _abl15 _np6; do
    timeout 120 ./b9s_probe$v $shape 2>&1 | grep -v "^M \|split3" >> $OUT
  done
done
grep -v "^   wave\|^M \|^rc 0\|shader clock" $OUT
