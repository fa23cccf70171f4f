#!/bin/bash
OUT=gpurun_out/${1:-s3}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
ABL_Q8S=1 timeout 300 ./tools/kernel_ablate > $OUT/ablate_q8s.txt 2>&1
cat $OUT/ablate_q8s.txt
timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_gpu_pipeline.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
LLAMAHIP_SKINNY=0 timeout 300 python tools/bench_ttft.py > $OUT/ttft_skinny0.json 2>$OUT/ttft.err
LLAMAHIP_SKINNY=1 timeout 300 python tools/bench_ttft.py > $OUT/ttft_skinny1.json 2>>$OUT/ttft.err
cat $OUT/ttft_skinny0.json $OUT/ttft_skinny1.json
