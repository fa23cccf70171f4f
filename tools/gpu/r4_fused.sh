#!/bin/bash
# round 4, GPU call 10: decode attention folded into the wq|wk|wv launch (LLAMAHIP_FUSED_ATTN=1) against the separate launch, same box:
# ids / logits hashes over 1000 steps x 3 runs (fp32, int8), tokens/s, then the decode-related GPU tests with the variable on
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4fused; mkdir -p $O
for f in 0 1 0 1; do
  LLAMAHIP_FUSED_ATTN=$f timeout 240 python tools/check_fused_attn.py --steps 1000 --runs 3 >> $O/ab_f32.jsonl 2>> $O/ab_f32.err; echo "f32 fused=$f rc=$?"
done
cat $O/ab_f32.jsonl
for f in 0 1 0 1; do
  LLAMAHIP_FUSED_ATTN=$f timeout 240 python tools/check_fused_attn.py --int8 --steps 1000 --runs 3 >> $O/ab_q8.jsonl 2>> $O/ab_q8.err; echo "q8 fused=$f rc=$?"
done
cat $O/ab_q8.jsonl
LLAMAHIP_FUSED_ATTN=1 timeout 600 python -m pytest tests/test_gpu_batch.py tests/test_context_swap.py tests/test_gpu_sample.py -m gpu -q -x > $O/tests_fused.log 2>&1; echo "tests fused rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests_fused.log | tail -6
