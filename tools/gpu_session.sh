#!/bin/bash
# One gpurun call = one session: full GPU test-suite, the default bench line, then A/B runs of kernel variants selected by
# environment switches (csrc/plan.hip: LLAMAHIP_GEMV_SA, LLAMAHIP_Q8_KERNEL, LLAMAHIP_Q8_WGPCU).  Output under gpurun_out/$1/.
OUT=gpurun_out/${1:-s}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
  tail -5 $OUT/pytest.log
fi
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
Q="--no-cpu-baseline --no-prefill"
LLAMAHIP_GEMV_SA=0 timeout 300 python bench.py $Q > $OUT/bench_f32_sa0.json 2>> $OUT/ab.err
LLAMAHIP_GEMV_SA=1 timeout 300 python bench.py $Q > $OUT/bench_f32_sa1.json 2>> $OUT/ab.err
for k in 0 1; do for w in 1 2; do
  LLAMAHIP_Q8_KERNEL=$k LLAMAHIP_Q8_WGPCU=$w timeout 300 python bench.py --int8 $Q > $OUT/bench_q8_k${k}_w${w}.json 2>> $OUT/ab.err
done; done
for f in $OUT/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d["roofline_token"]["frac_of_hbm_roofline"], d["roofline"]["kernel"], d["roofline"]["frac"], d.get("parity",{}).get("token_ids_match"), d.get("parity",{}).get("steps_compared"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
