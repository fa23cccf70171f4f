#!/bin/bash
# round 4, GPU call 9: sharded-pipeline context swap, one column tile on the LDS-DMA kernel (probe), long-context decode
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4c9; mkdir -p $O
timeout 900 python -m pytest tests/test_context_swap.py tests/test_gpu_pipeline.py -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -8
timeout 300 python tools/bench_ttft.py --ns 8,9,12,16,17 --reps 5 > $O/ttft_default.json 2> $O/ttft_default.err; echo "ttft rc=$?"; cat $O/ttft_default.json
LLAMAHIP_DMA_NCT1=1 timeout 300 python tools/bench_ttft.py --ns 9,12,16 --reps 5 > $O/ttft_nct1.json 2> $O/ttft_nct1.err; echo "ttft nct1 rc=$?"; cat $O/ttft_nct1.json
timeout 300 python tools/bench_pods.py --pods 9,12,16 --steps 32 > $O/pods_default.json 2> $O/pods_default.err; echo "pods rc=$?"; cat $O/pods_default.json
LLAMAHIP_DMA_NCT1=1 timeout 300 python tools/bench_pods.py --pods 9,12,16 --steps 32 > $O/pods_nct1.json 2> $O/pods_nct1.err; echo "pods nct1 rc=$?"; cat $O/pods_nct1.json
LLAMAHIP_DMA_NCT1=1 timeout 600 python -m pytest tests/test_gpu_llama.py tests/test_gpu_batch.py -m gpu -q -k "prefill_mfma_path or 7b_shape_slice or batched_decode_equals" > $O/tests_nct1.log 2>&1; echo "tests nct1 rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_nct1.log | tail -6
timeout 600 python tools/bench_longctx.py > $O/longctx.txt 2> $O/longctx.err; echo "longctx rc=$?"; tail -12 $O/longctx.txt
