#!/bin/bash
# round 4, GPU call 5: block-int8 on the LDS-DMA structure (k_stream_q8), embeddings (LH_T_OUTPUT), rows-kernel U probe
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4c5; mkdir -p $O
C=tools/stream_mm_check
{
echo "### checker: k_stream_q8 (mode 5) vs k_stream_mm2 int8 (mode 3); K = 512 checked"
for shape in "22016 512" "12288 512" "4096 512" "4096 1024"; do for n in 3 8 16 17 32 48; do for kc in 256 128; do
  echo "--- $shape n=$n KC=$kc"; timeout 120 $C $shape $n $kc 5 2>&1 | grep -E "k_stream_q8|max abs|do not fit|first wrong|HIP error"
done; done; done
for n in 8 24 40; do echo "--- 4096 1024 n=$n KC=256 K-split 2"; timeout 120 $C 4096 1024 $n 256 5 2 2>&1 | grep -E "k_stream_q8|max abs|do not fit|first wrong|HIP error"; done
export STREAM_CHECK_SKIP=1
echo "### timing (K full)"
for round in 1 2; do for shape in "22016 4096" "12288 4096" "4096 11008" "4096 4096"; do for n in 8 16 32 48; do
  echo "--- round $round shape $shape n=$n"
  timeout 60 $C $shape $n 128 3 2>&1 | grep -E "us per launch"
  for kc in 256 128; do timeout 60 $C $shape $n $kc 5 2>&1 | grep -E "us per launch|do not fit"; done
done; done; done
unset STREAM_CHECK_SKIP
} > $O/checker.log 2>&1
tail -4 $O/checker.log
timeout 600 python -m pytest tests/test_gpu_llama.py tests/test_gpu_batch.py tests/test_context_swap.py -m gpu -q -k "int8 or True or embeddings or q8" > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -12
timeout 300 python tools/bench_ttft.py --int8 --ns 1,2,3,4,8,16,24,32,48 --reps 5 > $O/ttft_q8.json 2> $O/ttft_q8.err; echo "ttft int8 rc=$?"; cat $O/ttft_q8.json
LLAMAHIP_STREAM_V=-1 timeout 300 python tools/bench_ttft.py --int8 --ns 3,8,16,32,48 --reps 5 > $O/ttft_q8_r3.json 2> $O/ttft_q8_r3.err; echo "ttft int8 round-3 kernel rc=$?"; cat $O/ttft_q8_r3.json
LLAMAHIP_Q8_KC=128 timeout 300 python tools/bench_ttft.py --int8 --ns 3,8,16,32,48 --reps 5 > $O/ttft_q8_kc128.json 2> $O/ttft_q8_kc128.err; echo "ttft int8 kc128 rc=$?"; cat $O/ttft_q8_kc128.json
timeout 300 python tools/bench_pods.py --int8 --pods 1,2,3,4,8,16,32,48 --steps 32 > $O/pods_q8.json 2> $O/pods_q8.err; echo "pods int8 rc=$?"; cat $O/pods_q8.json
for u in 3 4; do LLAMAHIP_ROWS_U=$u timeout 300 python tools/bench_pods.py --pods 5,8 --steps 32 > $O/pods_u$u.json 2> $O/pods_u$u.err; echo "pods U=$u rc=$?"; cat $O/pods_u$u.json; done
BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_2ranks_shared.json 2> $O/bench_2ranks_shared.err; echo "bench 2 ranks shared rc=$?"; python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r4c5/bench_2ranks_shared.json').read().strip().splitlines()[-1])
    print({k: d.get(k) for k in ('value','parity','pipeline_breakdown','single_stream')})
except Exception as e:
    print('parse failed', e); print(open('gpurun_out/r4c5/bench_2ranks_shared.err').read()[-1500:])
PY
