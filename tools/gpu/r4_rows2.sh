#!/bin/bash
# round 4, GPU call 15: rows kernels with packed FMAs over activation-row pairs (fp32) and the joint lane reduction (fp32 and int8):
# bit-identity / parity tests, then pods and prompts against the numbers of the final session
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4rows2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_llama.py tests/test_context_swap.py -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -6
timeout 300 python tools/bench_pods.py --pods 1,2,4,5,6,8 --steps 32 > $O/pods_f32.json 2> $O/pods_f32.err; echo "pods f32 rc=$?"
timeout 300 python tools/bench_pods.py --int8 --pods 1,2,3,4 --steps 32 > $O/pods_q8.json 2> $O/pods_q8.err; echo "pods q8 rc=$?"
timeout 300 python tools/bench_ttft.py --ns 2,3,4,5,6,8 --reps 5 > $O/ttft.json 2> $O/ttft.err; echo "ttft rc=$?"
python - <<'PY'
import json
for f in ('pods_f32','pods_q8'):
    d=json.load(open(f'gpurun_out/r4rows2/{f}.json')); print(f,{k:(v['tokens_per_s'],v['ms_per_step'],v['ids_equal_single_stream']) for k,v in d['by_pods'].items()})
print(open('gpurun_out/r4rows2/ttft.json').read())
PY
