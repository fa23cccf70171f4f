#!/bin/bash
# round 4, GPU call 30: the packed-multiply conversion as the product's k_stream_q8: int8 batch / prompt tests, pods
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4q8cvt3b; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_batch.py tests/test_context_swap.py -m gpu -q -x -k "int8 or True-" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout 200 python tools/bench_pods.py --int8 --pods 8,16,32,48,64 --steps 32 > $O/pods_q8.json 2> $O/pods_q8.err; echo "pods rc=$?"
python -c "
import json; d=json.load(open('$O/pods_q8.json')); print({k:(v['tokens_per_s'],v['ms_per_step'],v['ids_equal_single_stream']) for k,v in d['by_pods'].items()})"
timeout 200 python tools/bench_ttft.py --int8 --ns 8,16,32,48 --reps 5 2> $O/ttft.err | tee $O/ttft.json
