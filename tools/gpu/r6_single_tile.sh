#!/bin/bash
# round 6: wo / w2 of 7B at 17..48 rows WITHOUT the K-split: one 16-row tile per workgroup, 128-column chunks, 4-5 images (the DMA pattern probe says the loader alone then
# streams at 5.7 / 4.3 TB/s) against the shipped K-split pairs (64-column chunks, 4 images, ksplit 4 + reduce).  tools/stream_mm_check (tools/build_probes.sh), timing only.
cd "$(dirname "$0")/.." || exit 1
OUT=../gpurun_out/r6_single_tile.txt
: > $OUT
export STREAM_CHECK_SKIP=1
for shape in "4096 11008" "4096 4096"; do
  for n in 17 32 48; do
    echo "=== $shape n=$n" >> $OUT
    for img in 3 4 5; do STREAM_DMA_IMAGES=$img timeout 60 ./stream_mm_check $shape $n 128 4 >> $OUT 2>&1; done
    for img in 4 5; do STREAM_DMA_IMAGES=$img timeout 60 ./stream_mm_check $shape $n 64 4 >> $OUT 2>&1; done
    STREAM_DMA_IMAGES=4 STREAM_DMA_PIPE=1 timeout 60 ./stream_mm_check $shape $n 64 4 4 >> $OUT 2>&1
  done
done
grep -v "^max abs\|^M \|^$" $OUT
