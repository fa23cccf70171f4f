#!/bin/bash
# round 4, GPU call 24: the packed-FMA rows kernel with 2 / 3 / 4 weight rows in flight per wave (K = 4096 launches)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4rows5; mkdir -p $O
for u in 2 3 4; do
  LLAMAHIP_ROWS_U=$u timeout 200 python tools/bench_pods.py --pods 2,4,8 --steps 32 > $O/pods_u$u.json 2> $O/pods_u$u.err; echo "u=$u rc=$?"
  python -c "
import json; d=json.load(open('$O/pods_u$u.json')); print($u, {k:(v['tokens_per_s'],v['ms_per_step'],v['ids_equal_single_stream']) for k,v in d['by_pods'].items()})"
done
