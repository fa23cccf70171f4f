#!/bin/bash
# round 4, GPU call 3: eight column tiles (97..128 rows), the decode stream with eight activation rows, per-shape variants, effective clock
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4c3; mkdir -p $O
C=tools/stream_mm_check
{
echo "### checker: eight column tiles (K = 512 checked)"
for shape in "22016 512" "12288 512" "4096 512"; do for n in 97 112 128; do for pipe in 0 1; do
  echo "--- $shape n=$n pipe=$pipe"; STREAM_DMA_PIPE=$pipe STREAM_DMA_IMAGES=2 timeout 120 $C $shape $n 64 4 2>&1 | grep -E "k_stream_dma|max abs|do not fit|first wrong|HIP error"
done; done; done
for n in 97 128; do echo "--- 4096 1024 n=$n K-split 2"; STREAM_DMA_PIPE=1 STREAM_DMA_IMAGES=4 timeout 120 $C 4096 1024 $n 64 4 2 2>&1 | grep -E "k_stream_dma|max abs|do not fit|first wrong|HIP error|reduce"; done
export STREAM_CHECK_SKIP=1
echo "### timing (K full), effective clock"
for round in 1 2; do for shape in "22016 4096" "12288 4096" "4096 11008" "4096 4096"; do
  for n in 128; do for pipe in 0 1; do for img in 2 3 4; do echo "--- round $round shape $shape n=$n pipe=$pipe images=$img"; STREAM_DMA_PIPE=$pipe STREAM_DMA_IMAGES=$img timeout 60 $C $shape $n 64 4 2>&1 | grep -E "us per launch|MFMA wave|do not fit"; done; done; done
  for n in 32 64 96; do echo "--- round $round shape $shape n=$n"; STREAM_DMA_IMAGES=3 timeout 60 $C $shape $n 64 4 2>&1 | grep -E "us per launch|MFMA wave|do not fit"; done
done; done
for shape in "4096 11008" "4096 4096"; do for n in 128; do echo "--- K-split pairs: shape $shape n=$n"; for img in 3 4; do STREAM_DMA_PIPE=1 STREAM_DMA_IMAGES=$img timeout 60 $C $shape $n 64 4 2 2>&1 | grep -E "us per|reduce|MFMA wave"; done; done; done
unset STREAM_CHECK_SKIP
} > $O/checker.log 2>&1
tail -4 $O/checker.log
timeout 300 python tools/bench_ttft.py --ns 2,3,4,5,6,8,9,12,16,17,24,32,40,48,64,80,96,97,112,128 --reps 5 > $O/ttft_default.json 2> $O/ttft_default.err; echo "ttft default rc=$?"; cat $O/ttft_default.json
LLAMAHIP_ROWS_MAX=4 LLAMAHIP_STREAM_V=-1 timeout 300 python tools/bench_ttft.py --ns 5,6,8,32,64,97,112,128 --reps 5 > $O/ttft_r3.json 2> $O/ttft_r3.err; echo "ttft round-3 paths rc=$?"; cat $O/ttft_r3.json
timeout 300 python tools/bench_pods.py --pods 1,4,5,6,8,16,32,64 --steps 32 > $O/pods_default.json 2> $O/pods_default.err; echo "pods default rc=$?"; cat $O/pods_default.json
LLAMAHIP_ROWS_MAX=4 timeout 300 python tools/bench_pods.py --pods 5,6,8 --steps 32 > $O/pods_rows4.json 2> $O/pods_rows4.err; echo "pods rows<=4 rc=$?"; cat $O/pods_rows4.json
timeout 1200 python -m pytest tests/test_gpu_llama.py tests/test_gpu_batch.py tests/test_context_swap.py -m gpu -q -k "7b_shape_slice or batched_decode_equals or ticks_ or batches_come or prefill_mfma_path or bitwise or odd_shapes or long_context or pipeline_groups or swap or windows_tokens or argument_errors" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -15
