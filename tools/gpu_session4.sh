#!/bin/bash
OUT=gpurun_out/${1:-s4}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_llama.py tests/test_gpu_pipeline.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
LLAMAHIP_SKINNY=0 timeout 300 python tools/bench_ttft.py --ns 1,2,4,8,9 > $OUT/ttft_skinny0.json 2>$OUT/ttft.err
LLAMAHIP_SKINNY=1 timeout 300 python tools/bench_ttft.py --ns 1,2,4,8,9 > $OUT/ttft_skinny1.json 2>>$OUT/ttft.err
cat $OUT/ttft_skinny0.json $OUT/ttft_skinny1.json
timeout 300 python bench.py --int8 --no-cpu-baseline --no-prefill > $OUT/bench_q8.json 2>> $OUT/ab.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/s4/bench_q8.json").read().strip().splitlines()[-1])
print("int8", d["value"], d["roofline_token"]["frac_of_hbm_roofline"]); print(d["kernels"])
PY
