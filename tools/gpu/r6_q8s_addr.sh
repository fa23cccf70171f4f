mkdir -p gpurun_out/r6s
timeout 200 ./tools/q8s_phase_probe > gpurun_out/r6s/q8s_phase_final.txt 2>&1
python -m pytest tests -x -q -m gpu -k "int8 or q8 or headline or batch or reproducible or pipeline_groups" 2>&1 | tail -5 > gpurun_out/r6s/tests.log
python bench.py --no-cpu-baseline > gpurun_out/r6s/bench.json 2> gpurun_out/r6s/bench.err
cat gpurun_out/r6s/tests.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r6s/bench.json'))
print(d['value'], d['int8_decode']['tokens_per_s'], d['int8_decode']['kernels'], d['int8_decode']['pods_batched'])
PY
