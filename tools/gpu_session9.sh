#!/bin/bash
# resident decode kernel: its tests, the bench line with it on and off, the traced standalone probe
OUT=gpurun_out/${1:-s9}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_llama.py -m gpu -x -q -k "resident or greedy_decode or concurrent_pods or pods_come" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
LLAMAHIP_RESIDENT=1 timeout 300 python bench.py --no-cpu-baseline --no-prefill > $OUT/bench_resident1.json 2> $OUT/bench_resident1.err; echo "bench1 rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-prefill > $OUT/bench_resident0.json 2> $OUT/bench_resident0.err; echo "bench0 rc=$?"
timeout 60 ./tools/resident_probe > $OUT/resident_trace.txt 2>&1
python - "$OUT" <<'PY'
import json,sys
o=sys.argv[1]
for n in ("bench_resident1","bench_resident0"):
    try:
        d=json.loads(open(o+"/"+n+".json").read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["tokens_stream0"][:4])
    except Exception as e: print(n, "ERR", e)
PY
cat $OUT/resident_trace.txt
