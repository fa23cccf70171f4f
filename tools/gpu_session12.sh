#!/bin/bash
OUT=gpurun_out/${1:-s12}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --no-cpu-baseline --no-prefill > $OUT/bench_quick.json 2>> $OUT/ab.err
timeout 300 python tools/bench_pipeline_overhead.py > $OUT/pipeline_overhead.json 2>> $OUT/ab.err; cat $OUT/pipeline_overhead.json
python - "$OUT" <<'PY'
import json,sys
o=sys.argv[1]
for f in ("bench_default","bench_quick"):
    try:
        d=json.loads(open(o+"/"+f+".json").read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d["roofline_token"]["frac_of_hbm_roofline"], d["roofline"]["frac"], d.get("parity",{}).get("token_ids_match"), d.get("sampled_decode",{}).get("tokens_per_s"), d.get("int8_decode",{}).get("tokens_per_s"), d.get("prompt_8_tokens",{}).get("ms"))
    except Exception as e: print(f, "ERR", e)
PY
