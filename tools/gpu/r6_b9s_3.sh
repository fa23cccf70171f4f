#!/bin/bash
# round 6: what bounds k_stream_b9 - timing-only ablation builds of tools/b9s_probe (results wrong by construction)
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_b9s_3.txt
: > $OUT
export B9S_SKIP_CHECK=1 B9S_NO_DMA=1
for shape in "11008 4096 64 2 2 1" "11008 4096 32 2 2 1" "4096 4096 64 2 3" "4096 11008 64 2 1 0 4" "11008 4096 128 2 2 1"; do
  echo "==== $shape" >> $OUT
  for v in "" _heavy _abl1 _abl2 _abl4 _abl8 _abl9 _abl15 _np6; do
    timeout 120 ./b9s_probe$v $shape 2>&1 | grep -v "^M \|split3" >> $OUT
  done
done
cat $OUT
