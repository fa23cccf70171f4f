#!/bin/bash
# round 6, evidence session: the full GPU suite, the bench line, kernel traces + counters of the final build, side measurements of configs 3 / 4 / 5,
# the 8- and 2-rank rehearsal on one GPU.  The pmc:f32 step regenerates the HBM-traffic file bench.py reads (stamped with the kernel-source hash):
# copy gpurun_out/${R6NAME:-r6final}/pmc_f32.json to profiles/pmc_traffic.json afterwards.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
bash tools/gpu_run.sh ${R6NAME:-r6final} \
  tests \
  bench \
  "trace:f32:python bench.py --no-cpu-baseline --no-prefill" \
  "pmc:f32:FETCH_SIZE:python bench.py --no-cpu-baseline --no-prefill" \
  "trace:pods64:python tools/bench_pods.py --pods 64 --steps 16" \
  "pmc:pods64:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE:python tools/bench_pods.py --pods 64 --steps 8" \
  "pmc:pods64f:FETCH_SIZE:python tools/bench_pods.py --pods 64 --steps 8" \
  "trace:q8:python bench.py --int8 --no-cpu-baseline --no-prefill" \
  "trace:p13:python tools/bench_prefill.py --shape 13B --n 1024 --layers 12" \
  "pmc:p13:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE:python tools/bench_prefill.py --shape 13B --n 1024 --layers 6 --reps 1" \
  "sh:prefill13:python tools/bench_prefill.py --shape 13B --n 1024 --reps 5; python tools/bench_prefill.py --shape 13B --n 1024 --reps 3 --int8" \
  "sh:shard65:python tools/bench_65b_shard.py" \
  "sh:ttft:python tools/bench_ttft.py --ns 1,2,4,8,9,16,17,32,48,49,56,64,65,96,128,129,160,192,193,256,512,1024 --reps 3" \
  "sh:ttft8:python tools/bench_ttft.py --int8 --ns 1,2,4,8,16,32,48,64,96,128 --reps 3" \
  "sh:pods:python tools/bench_pods.py --steps 24" \
  "sh:pods8:python tools/bench_pods.py --int8 --steps 24" \
  "sh:host:LLAMAGO_TIMING=1 LLAMAHIP_TIMING=1 python tools/host_timing_probe.py" \
  "sh:bench8:BENCH_SHARED_GPU=1 python bench.py --gpus 8 --steps 8 --warmup 2 --no-cpu-baseline" \
  "sh:bench2:BENCH_SHARED_GPU=1 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline"
python tools/pmc_dump.py "$(find gpurun_out/${R6NAME:-r6final}/pmc_p13 -name '*.db' | head -1)" k_ > gpurun_out/${R6NAME:-r6final}/pmc_p13_dump.txt 2>&1
python tools/pmc_dump.py "$(find gpurun_out/${R6NAME:-r6final}/pmc_pods64 -name '*.db' | head -1)" k_stream > gpurun_out/${R6NAME:-r6final}/pmc_pods64_dump.txt 2>&1
python tools/pmc_dump.py "$(find gpurun_out/${R6NAME:-r6final}/pmc_pods64f -name '*.db' | head -1)" k_stream > gpurun_out/${R6NAME:-r6final}/pmc_pods64f_dump.txt 2>&1
find gpurun_out/${R6NAME:-r6final} -name "*.db" -size +8M -delete; find gpurun_out/${R6NAME:-r6final} -name "*.csv" -size +4M -delete
du -sh gpurun_out/${R6NAME:-r6final}
