#!/bin/bash
# round 4, GPU call 14: block-int8 rows kernel (2..4 activation rows): 256 against 512 threads per workgroup, and where its wave cycles go
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4q8rows; mkdir -p $O
for t in 256 512 256 512; do
  LLAMAHIP_Q8R_TH=$t timeout 200 python tools/bench_pods.py --int8 --pods 2,3,4 --steps 32 >> $O/pods_th.jsonl 2>> $O/pods_th.err; echo "th=$t rc=$?"
done
python - <<'PY'
import json
for l in open('gpurun_out/r4q8rows/pods_th.jsonl'):
    d=json.loads(l); print({k:(v['tokens_per_s'],v['ms_per_step']) for k,v in d['by_pods'].items()})
PY
bash tools/gpu_run.sh r4q8rows "pmc:q8rows:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES:python tools/bench_pods.py --int8 --pods 1,2,4 --steps 8"
python tools/pmc_dump.py "$(find gpurun_out/r4q8rows/pmc_q8rows -name '*.db' | head -1)" k_gemv_q8 > $O/pmc_dump.txt; head -60 $O/pmc_dump.txt
