#!/bin/bash
# round 4, GPU call 16: the shader clock under fp32 MFMA load on every SIMD (tools/mfma_clock_probe), and bench.py --gpus 3 with all ranks on the one GPU
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4clock3; mkdir -p $O
timeout 60 tools/mfma_clock_probe 200000 1 > $O/mfma_clock.txt 2>&1; echo "probe rc=$?"
timeout 60 tools/mfma_clock_probe 100000 2 >> $O/mfma_clock.txt 2>&1; echo "probe2 rc=$?"
cat $O/mfma_clock.txt
BENCH_SHARED_GPU=1 timeout 400 python bench.py --gpus 3 --steps 8 --warmup 2 > $O/bench_3ranks.json 2> $O/bench_3ranks.err; echo "bench3 rc=$?"; tail -c 1500 $O/bench_3ranks.json; tail -3 $O/bench_3ranks.err
