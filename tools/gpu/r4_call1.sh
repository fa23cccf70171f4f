#!/bin/bash
# round 4, GPU call 1: the LDS-DMA loader probe (tools/kernels_stream_dma.h, mode 4) against k_stream_mm2 (mode 2)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4c1; mkdir -p $O
C=tools/stream_mm_check
{
echo "### correctness (K = 512, checked against the float64 host product)"
for shape in "22016 512" "12288 512" "4096 512"; do
  for n in 17 32 48 64; do
    for img in 2 3 4; do echo "--- $shape n=$n KC=64 images=$img"; STREAM_DMA_IMAGES=$img timeout 120 $C $shape $n 64 4 2>&1 | grep -E "k_stream_dma|max abs|do not fit|first wrong|HIP error"; done
    [ $n -le 48 ] && { echo "--- $shape n=$n KC=128 images=2"; STREAM_DMA_IMAGES=2 timeout 120 $C $shape $n 128 4 2>&1 | grep -E "k_stream_dma|max abs|do not fit|first wrong|HIP error"; }
  done
done
} > $O/dma_correct.log 2>&1
{
echo "### timing (K full; STREAM_CHECK_SKIP), three interleaved rounds"
export STREAM_CHECK_SKIP=1
for round in 1 2 3; do
for shape in "22016 4096" "12288 4096" "4096 11008" "4096 4096"; do
  for n in 32 48 64; do
    echo "--- round $round shape $shape n=$n"
    timeout 60 $C $shape $n 64 2 2>&1 | grep -E "us per launch"
    [ $n -le 48 ] && timeout 60 $C $shape $n 128 2 2>&1 | grep -E "us per launch"
    for img in 2 3 4; do STREAM_DMA_IMAGES=$img timeout 60 $C $shape $n 64 4 2>&1 | grep -E "us per launch|do not fit"; done
    [ $n -le 48 ] && STREAM_DMA_IMAGES=2 timeout 60 $C $shape $n 128 4 2>&1 | grep -E "us per launch|do not fit"
  done
done
done
} > $O/dma_timing.log 2>&1
tail -5 $O/dma_correct.log; tail -30 $O/dma_timing.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.json
