#!/bin/bash
# round 4, GPU call 22 (round-robin chains): v_pk_fma_f32 issue rate; the fp32 rows kernel with packed FMAs over activation-row pairs and INDEPENDENT six-step lane reductions
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4rows4; mkdir -p $O
timeout 60 tools/pk_fma_probe > $O/pk_probe.txt 2>&1; cat $O/pk_probe.txt
timeout 400 python -m pytest tests/test_gpu_batch.py -m gpu -q -x -k "bit_identical or neighbours or odd_shapes" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout 300 python tools/bench_pods.py --pods 1,2,4,5,6,8 --steps 32 > $O/pods_f32.json 2> $O/pods_f32.err; echo "pods f32 rc=$?"
timeout 300 python tools/bench_ttft.py --ns 2,4,5,8 --reps 5 > $O/ttft.json 2> $O/ttft.err; echo "ttft rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4rows4/pods_f32.json')); print({k:(v['tokens_per_s'],v['ms_per_step'],v['ids_equal_single_stream']) for k,v in d['by_pods'].items()})
print(open('gpurun_out/r4rows4/ttft.json').read())
PY
