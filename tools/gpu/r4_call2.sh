#!/bin/bash
# round 4, GPU call 2: the product k_stream_dma (pipelined operands), variants A/B in the model, then the GPU suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4c2; mkdir -p $O
C=tools/stream_mm_check
{
echo "### checker: product kernel, pipelined operands (K = 512 checked; K = 4096 timed)"
for shape in "22016 512" "12288 512" "4096 512"; do for n in 17 32 48 64 96; do for img in 3 4; do
  echo "--- $shape n=$n KC=64 images=$img pipe"; STREAM_DMA_PIPE=1 STREAM_DMA_IMAGES=$img timeout 120 $C $shape $n 64 4 2>&1 | grep -E "k_stream_dma|max abs|do not fit|first wrong|HIP error"
done; done; done
for n in 32 48; do echo "--- 22016 512 n=$n KC=128 images=2 pipe"; STREAM_DMA_PIPE=1 STREAM_DMA_IMAGES=2 timeout 120 $C 22016 512 $n 128 4 2>&1 | grep -E "k_stream_dma|max abs|do not fit|first wrong|HIP error"; done
for n in 32 48 64; do echo "--- 4096 1024 n=$n KC=64 images=3 pipe K-split 2"; STREAM_DMA_PIPE=1 STREAM_DMA_IMAGES=3 timeout 120 $C 4096 1024 $n 64 4 2 2>&1 | grep -E "k_stream_dma|max abs|do not fit|first wrong|HIP error|reduce"; done
export STREAM_CHECK_SKIP=1
for round in 1 2; do for shape in "22016 4096" "12288 4096" "4096 11008" "4096 4096"; do for n in 32 48 64 96; do
  echo "--- round $round shape $shape n=$n"
  for pipe in 0 1; do for img in 3 4; do STREAM_DMA_PIPE=$pipe STREAM_DMA_IMAGES=$img timeout 60 $C $shape $n 64 4 2>&1 | grep -E "us per launch|do not fit|registers"; done; done
  [ $n -le 48 ] && for pipe in 0 1; do STREAM_DMA_PIPE=$pipe STREAM_DMA_IMAGES=2 timeout 60 $C $shape $n 128 4 2>&1 | grep -E "us per launch|do not fit|registers"; done
done; done; done
for shape in "4096 11008" "4096 4096"; do for n in 32 48 64; do
  echo "--- K-split pairs: shape $shape n=$n"
  timeout 60 $C $shape $n 64 2 2 2>&1 | grep -E "us per|reduce"
  for img in 3 4; do STREAM_DMA_PIPE=1 STREAM_DMA_IMAGES=$img timeout 60 $C $shape $n 64 4 2 2>&1 | grep -E "us per|reduce"; done
done; done
unset STREAM_CHECK_SKIP
} > $O/checker.log 2>&1
tail -3 $O/checker.log
for v in -1 0 1 2 3 4; do
  LLAMAHIP_STREAM_V=$v timeout 300 python tools/bench_ttft.py --ns 17,24,32,40,48,56,64,80,96 --reps 5 > $O/ttft_v$v.json 2> $O/ttft_v$v.err; echo "ttft v=$v rc=$?"; cat $O/ttft_v$v.json
done
for v in -1 1; do
  LLAMAHIP_STREAM_V=$v timeout 300 python tools/bench_pods.py --pods 17,32,48,64 --steps 24 > $O/pods_v$v.json 2> $O/pods_v$v.err; echo "pods v=$v rc=$?"; cat $O/pods_v$v.json
done
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_gpu_llama.py tests/test_gpu_batch.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -5
