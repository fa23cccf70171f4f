#!/bin/bash
# round 5, evidence session: the bench line, kernel traces + counters of the final build, configs 3 / 4 / 5 side measurements
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
bash tools/gpu_run.sh r5final \
  bench \
  "trace:f32:python bench.py --no-cpu-baseline --no-prefill" \
  "pmc:f32:FETCH_SIZE:python bench.py --no-cpu-baseline --no-prefill" \
  "trace:q8:python bench.py --int8 --no-cpu-baseline --no-prefill" \
  "trace:pods16q8:python tools/bench_pods.py --int8 --pods 16 --steps 16" \
  "trace:p13:python tools/bench_prefill.py --shape 13B --n 1024 --layers 12" \
  "pmc:p13:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE:python tools/bench_prefill.py --shape 13B --n 1024 --layers 6 --reps 1" \
  "sh:prefill13:python tools/bench_prefill.py --shape 13B --n 1024 --reps 5; python tools/bench_prefill.py --shape 13B --n 1024 --reps 3 --int8" \
  "sh:shard65:python tools/bench_65b_shard.py" \
  "sh:ttft:python tools/bench_ttft.py --ns 1,2,4,8,9,16,17,32,48,64,96,128 --reps 3" \
  "sh:ttft8:python tools/bench_ttft.py --int8 --ns 1,2,4,8,9,16,17,32,48,64,65,96,128 --reps 3" \
  "sh:pods:python tools/bench_pods.py --steps 24" \
  "sh:pods8:python tools/bench_pods.py --int8 --steps 24" \
  "sh:host:LLAMAGO_TIMING=1 LLAMAHIP_TIMING=1 python tools/host_timing_probe.py"
python tools/pmc_dump.py "$(find gpurun_out/r5final/pmc_p13 -name '*.db' | head -1)" k_ > gpurun_out/r5final/pmc_p13_dump.txt 2>&1; head -30 gpurun_out/r5final/pmc_p13_dump.txt
find gpurun_out/r5final -name "*.db" -size +8M -delete; find gpurun_out/r5final -name "*.csv" -size +4M -delete
du -sh gpurun_out/r5final
