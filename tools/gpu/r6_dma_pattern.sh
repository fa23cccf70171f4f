#!/bin/bash
# round 6: tools/dma_pattern_probe on the four matrices of a 7B layer (build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dma_pattern_probe tools/dma_pattern_probe.hip)
cd "$(dirname "$0")/.." || exit 1
mkdir -p ../gpurun_out
OUT=../gpurun_out/r6_dma_pattern.txt
: > $OUT
for s in "22016 4096" "12288 4096" "4096 11008" "4096 4096"; do timeout 120 ./dma_pattern_probe $s >> $OUT 2>&1; echo "rc $?" >> $OUT; done
cat $OUT
