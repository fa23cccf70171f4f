#!/bin/bash
# round 4, GPU call 8: k_stream_q8 conversion forms (unsigned cvt + fma | sign-extending SDWA cvt + mul | byte permute), then the int8 model numbers
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4c8; mkdir -p $O
{
echo "### checked (K = 512): the default build (SDWA convert + multiply)"
for shape in "22016 512" "12288 512" "4096 1024"; do for n in 3 16 32 48 64; do echo "--- $shape n=$n"; timeout 120 tools/stream_mm_check $shape $n 256 5 2>&1 | grep -E "k_stream_q8|max abs|do not fit|first wrong|HIP error"; done; done
export STREAM_CHECK_SKIP=1
for round in 1 2; do for shape in "22016 4096" "12288 4096" "4096 11008" "4096 4096"; do for n in 16 32 48; do
  echo "--- round $round shape $shape n=$n"
  for b in "" _q8cvt0 _q8cvt2; do for kc in 256 128; do echo -n "build '$b' KC=$kc: "; timeout 60 tools/stream_mm_check$b $shape $n $kc 5 2>&1 | grep -E "us per launch|do not fit"; done; done
done; done; done
unset STREAM_CHECK_SKIP
} > $O/q8_cvt.log 2>&1
tail -8 $O/q8_cvt.log
timeout 300 python tools/bench_ttft.py --int8 --ns 1,2,3,4,8,16,24,32,48,64 --reps 5 > $O/ttft_q8.json 2> $O/ttft_q8.err; echo "ttft int8 rc=$?"; cat $O/ttft_q8.json
timeout 300 python tools/bench_pods.py --int8 --pods 1,2,4,8,16,32,48,64 --steps 32 > $O/pods_q8.json 2> $O/pods_q8.err; echo "pods int8 rc=$?"; cat $O/pods_q8.json
timeout 600 python -m pytest tests/test_gpu_llama.py tests/test_gpu_batch.py tests/test_context_swap.py -m gpu -q -k "int8 or True or q8" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -6
