#!/bin/bash
# round 6, third session's evidence run on the final tree: GPU suite, bench line, kernel traces (fp32 + block-int8), the HBM-traffic counters bench.py reads
# (copy gpurun_out/r6final3/pmc_f32.json to profiles/pmc_traffic.json afterwards), pods / prompts fp32 + int8, the 8- and 2-rank rehearsal on one GPU.
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
bash tools/gpu_run.sh ${R6NAME:-r6final3} \
  tests \
  bench \
  "trace:f32:python bench.py --no-cpu-baseline --no-prefill" \
  "pmc:f32:FETCH_SIZE:python bench.py --no-cpu-baseline --no-prefill" \
  "trace:q8:python bench.py --int8 --no-cpu-baseline --no-prefill" \
  "sh:pods:python tools/bench_pods.py --steps 24" \
  "sh:pods8:python tools/bench_pods.py --int8 --steps 24" \
  "sh:ttft:python tools/bench_ttft.py --ns 1,2,4,8,9,16,17,32,48,49,64,65,96,128,129,160,192,256 --reps 3" \
  "sh:ttft8:python tools/bench_ttft.py --int8 --ns 1,2,4,8,16,32,48,64,96,128 --reps 3" \
  "sh:prefill13:python tools/bench_prefill.py --shape 13B --n 1024 --reps 5; python tools/bench_prefill.py --shape 13B --n 1024 --reps 3 --int8" \
  "sh:longctx:python tools/bench_longctx.py" \
  "sh:bench8:BENCH_SHARED_GPU=1 python bench.py --gpus 8 --steps 8 --warmup 2 --no-cpu-baseline" \
  "sh:bench2:BENCH_SHARED_GPU=1 python bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline"
find gpurun_out/${R6NAME:-r6final3} -name "*.db" -size +8M -delete; find gpurun_out/${R6NAME:-r6final3} -name "*.csv" -size +4M -delete
du -sh gpurun_out/${R6NAME:-r6final3}
