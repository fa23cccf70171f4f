#!/bin/bash
# round 4, GPU call 28: block-int8 decode stream, lanes per weight row (K = 4096 launches): 256 (product) / 128 / 64, rows in flight per register set
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4q8tpr; mkdir -p $O
for cfg in "256 2" "128 2" "128 3" "64 1" "64 2" "256 2"; do
  set -- $cfg
  LLAMAHIP_Q8_TPR=$1 LLAMAHIP_Q8_U=$2 timeout 120 python tools/check_fused_attn.py --int8 --steps 400 --runs 2 2>> $O/err.txt | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('tpr $1 u $2', [(r['ids_sha'][:6], r['tok_s']) for r in d['runs']])"
done
