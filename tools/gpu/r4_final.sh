#!/bin/bash
# round 4, evidence session: the whole GPU suite, the bench line, kernel traces + counters of the final build
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
bash tools/gpu_run.sh r4final \
  tests \
  bench \
  "trace:f32:python bench.py --no-cpu-baseline --no-prefill" \
  "pmc:f32:FETCH_SIZE:python bench.py --no-cpu-baseline --no-prefill" \
  "trace:q8:python bench.py --int8 --no-cpu-baseline --no-prefill" \
  "pmc:stream:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE:python tools/bench_ttft.py --ns 32,48,64,96,128 --reps 2" \
  "trace:ttft:python tools/bench_ttft.py --ns 32,48,96 --reps 3" \
  "trace:pods16q8:python tools/bench_pods.py --int8 --pods 16 --steps 16"
python tools/pmc_dump.py "$(find gpurun_out/r4final/pmc_stream -name '*.db' | head -1)" k_stream > gpurun_out/r4final/pmc_stream_dump.txt 2>&1; head -40 gpurun_out/r4final/pmc_stream_dump.txt
