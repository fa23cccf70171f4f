#!/bin/bash
# round 4, GPU call 13: device-coherent loads for every row again, one arrival counter per 256 bytes, A/B on one box
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4fused4; mkdir -p $O
for f in 0 1 0 1; do
  LLAMAHIP_FUSED_ATTN=$f timeout 240 python tools/check_fused_attn.py --steps 1000 --runs 3 >> $O/ab_f32.jsonl 2>> $O/ab_f32.err; echo "f32 fused=$f rc=$?"
done
cat $O/ab_f32.jsonl
for f in 0 1; do
  LLAMAHIP_FUSED_ATTN=$f timeout 240 python tools/check_fused_attn.py --int8 --steps 1000 --runs 3 >> $O/ab_q8.jsonl 2>> $O/ab_q8.err; echo "q8 fused=$f rc=$?"
done
cat $O/ab_q8.jsonl
