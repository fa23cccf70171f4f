#!/bin/bash
# round 4, GPU call 6: k_stream_q8 with the conversion pipelined under the MFMAs; 64-row int8; the whole GPU suite
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4c6; mkdir -p $O
C=tools/stream_mm_check
{
echo "### checker: k_stream_q8 (mode 5); K = 512 / 1024 checked"
for shape in "22016 512" "12288 512" "4096 1024"; do for n in 3 16 17 32 48 64; do for kc in 256 128; do
  echo "--- $shape n=$n KC=$kc"; timeout 120 $C $shape $n $kc 5 2>&1 | grep -E "k_stream_q8|max abs|do not fit|first wrong|HIP error"
done; done; done
for n in 8 40 64; do echo "--- 4096 1024 n=$n KC=256 K-split 2"; timeout 120 $C 4096 1024 $n 256 5 2 2>&1 | grep -E "k_stream_q8|max abs|do not fit|first wrong|HIP error"; done
export STREAM_CHECK_SKIP=1
echo "### timing (K full)"
for round in 1 2; do for shape in "22016 4096" "12288 4096" "4096 11008" "4096 4096"; do for n in 8 16 32 48 64; do
  echo "--- round $round shape $shape n=$n"
  for kc in 256 128; do timeout 60 $C $shape $n $kc 5 2>&1 | grep -E "us per launch|do not fit"; done
done; done; done
unset STREAM_CHECK_SKIP
} > $O/checker.log 2>&1
tail -4 $O/checker.log
timeout 300 python tools/bench_ttft.py --int8 --ns 1,2,3,4,8,16,24,32,48,64 --reps 5 > $O/ttft_q8.json 2> $O/ttft_q8.err; echo "ttft int8 rc=$?"; cat $O/ttft_q8.json
LLAMAHIP_Q8_KC=128 timeout 300 python tools/bench_ttft.py --int8 --ns 3,8,16,32,48,64 --reps 5 > $O/ttft_q8_kc128.json 2> $O/ttft_q8_kc128.err; echo "ttft int8 kc128 rc=$?"; cat $O/ttft_q8_kc128.json
timeout 300 python tools/bench_pods.py --int8 --pods 1,2,3,4,8,16,32,48,64 --steps 32 > $O/pods_q8.json 2> $O/pods_q8.err; echo "pods int8 rc=$?"; cat $O/pods_q8.json
timeout 1800 python -m pytest tests -m gpu -q > $O/tests_all.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_all.log | tail -12
