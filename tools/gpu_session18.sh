#!/bin/bash
# K-split wo / w2 at 17..32 rows inside the model: parity subset, then TTFT with the split off / w2 only / both, fp32 and int8.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s18; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_llama.py tests/test_gpu_ops.py -m gpu -x -q -k "prefill or odd_shapes or chunked or reproducible or int8 or 7b_shape or mul_mat or stage or pipeline" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for k in 0 2 3 1; do
  echo "KSPLIT=$k fp32"; LLAMAHIP_STREAM_KSPLIT=$k timeout 300 python tools/bench_ttft.py --ns 16,17,24,32 2>/dev/null | tail -1
done | tee $O/ttft_ksplit.txt
for k in 0 3; do
  echo "KSPLIT=$k int8"; LLAMAHIP_STREAM_KSPLIT=$k timeout 300 python tools/bench_ttft.py --ns 24,32 --int8 2>/dev/null | tail -1
done | tee -a $O/ttft_ksplit.txt
