#!/bin/bash
# round 4, GPU call 4: context swap (resident loops, batch rows, unsharded pipeline), seven column tiles, rows kernel prologue / epilogue, traces
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4c4; mkdir -p $O
C=tools/stream_mm_check
{
for shape in "22016 512" "12288 512" "4096 512"; do for n in 97 112; do echo "--- $shape n=$n"; STREAM_DMA_IMAGES=3 timeout 120 $C $shape $n 64 4 2>&1 | grep -E "k_stream_dma|max abs|do not fit|first wrong|HIP error"; done; done
export STREAM_CHECK_SKIP=1
for shape in "22016 4096" "12288 4096" "4096 11008" "4096 4096"; do for n in 112; do echo "--- shape $shape n=$n"; STREAM_DMA_IMAGES=3 timeout 60 $C $shape $n 64 4 2>&1 | grep -E "us per launch|MFMA wave|do not fit"; done; done
unset STREAM_CHECK_SKIP
} > $O/checker.log 2>&1
tail -3 $O/checker.log
timeout 900 python -m pytest tests/test_context_swap.py tests/test_gpu_batch.py -m gpu -q > $O/tests_a.log 2>&1; echo "tests_a rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_a.log | tail -12
timeout 600 python -m pytest tests/test_gpu_llama.py -m gpu -q -k "7b_shape_slice or prefill_mfma_path or resident or concurrent or come_and_go" > $O/tests_b.log 2>&1; echo "tests_b rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_b.log | tail -8
timeout 300 python tools/bench_ttft.py --ns 2,4,5,6,8,16,32,64,96,97,112,128 --reps 5 > $O/ttft.json 2> $O/ttft.err; echo "ttft rc=$?"; cat $O/ttft.json
timeout 300 python tools/bench_pods.py --pods 1,4,5,6,8,16,32,64 --steps 32 > $O/pods.json 2> $O/pods.err; echo "pods rc=$?"; cat $O/pods.json
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$O/prof_pods8" -o pods8 -- bash -c "cd $OLDPWD && python tools/bench_pods.py --pods 8 --steps 16") > $O/trace_pods8.log 2>&1; echo "trace pods8 rc=$?"
python tools/prof_summary.py "$(find $O/prof_pods8 -name '*.db' | head -1)" > $O/trace_pods8.txt 2>&1; head -24 $O/trace_pods8.txt
find $O/prof_pods8 -name "*.csv" -size +20M -delete
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$O/prof_ttft64" -o ttft64 -- bash -c "cd $OLDPWD && python tools/bench_ttft.py --ns 64,128 --reps 3") > $O/trace_ttft64.log 2>&1; echo "trace ttft rc=$?"
python tools/prof_summary.py "$(find $O/prof_ttft64 -name '*.db' | head -1)" > $O/trace_ttft64.txt 2>&1; head -30 $O/trace_ttft64.txt
find $O/prof_ttft64 -name "*.csv" -size +20M -delete
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 400 $O/bench.json
