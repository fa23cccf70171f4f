#!/bin/bash
# round 6: k_stream_b9 standalone (tools/b9s_probe; build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/b9s_probe tools/b9s_probe.hip, and with
# -DB9S_ABLATE=8 as tools/b9s_probe_abl8 for the compute side alone) - correctness on ragged shapes, the 7B launches beside k_stream_dma
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_b9s_probe.txt
: > $OUT
run() { echo "== $*" >> $OUT; timeout 120 ./b9s_probe "$@" >> $OUT 2>&1; echo "rc $?" >> $OUT; }
run 352 1024 13 -
run 352 1024 33 -
run 352 1024 40 -
run 352 1024 64 - 2 1
run 2048 512 50 - 3
run 2048 512 64 - 3
run 4096 512 17 - 1 0 2
for n in 16 32 48 64; do
  run 11008 4096 $n - 2 1
  run 4096 4096 $n - 3
  run 4096 4096 $n - 1 0 4
  run 4096 11008 $n - 1 0 4
done
run 32000 4096 48 -
export B9S_SKIP_CHECK=1 B9S_NO_DMA=1
for shape in "11008 4096 64 - 2 1" "4096 4096 64 - 3" "4096 11008 64 - 1 0 4" "4096 4096 64 - 1 0 4"; do
  echo "==== $shape" >> $OUT
  timeout 120 ./b9s_probe_abl8 $shape 2>/dev/null 2>&1 | grep -v "^M \|split3" >> $OUT
done
grep -v "^M \|^rc 0" $OUT
