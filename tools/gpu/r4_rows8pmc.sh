#!/bin/bash
# round 4, GPU call 20: where the wave cycles of the eight-row fp32 rows kernel go, against the one-row decode stream (SQ counters)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4rows8pmc; mkdir -p $O
bash tools/gpu_run.sh r4rows8pmc "pmc:rows8:SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM:python tools/bench_pods.py --pods 1,4,8 --steps 8" > $O/run.log 2>&1
python tools/pmc_dump.py "$(find gpurun_out/r4rows8pmc/pmc_rows8 -name '*.db' | head -1)" k_gemv > $O/pmc_dump.txt; cat $O/pmc_dump.txt | head -150
rm -rf gpurun_out/r4rows8pmc/pmc_rows8
