#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_b9s_7.txt
: > $OUT
export B9S_SKIP_CHECK=1 B9S_NO_DMA=1
for rep in 1 2; do
for shape in "11008 4096 64 - 2 1" "4096 4096 64 - 3" "4096 11008 64 - 1 0 4"; do
  echo "==== $shape" >> $OUT
  for v in _bare0 _bare8 _bare16 _bare0_abl8 _bare8_abl8 _bare16_abl8; do
    timeout 120 ./b9s_probe$v $shape 2>&1 | grep -v "^M \|split3" >> $OUT
  done
done
done
cat $OUT
