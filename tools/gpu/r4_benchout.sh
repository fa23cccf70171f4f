#!/bin/bash
# round 4, GPU call 18: bench.py's stdout is exactly one JSON line (N = 1, and N = 2 / 3 with the ranks sharing the GPU: gloo's C++ side printed in front of it)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4benchout; mkdir -p $O
timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-prefill > $O/n1.out 2> $O/n1.err; echo "n1 rc=$? lines=$(wc -l < $O/n1.out)"; python -c "import json;d=json.load(open('$O/n1.out'));print(d['value'],d['roofline']['frac'])"
BENCH_SHARED_GPU=1 timeout 400 python bench.py --gpus 3 --steps 8 --warmup 2 > $O/n3.out 2> $O/n3.err; echo "n3 rc=$? lines=$(wc -l < $O/n3.out)"; python -c "import json;d=json.load(open('$O/n3.out'));print(d['value'],d['parity'],[(r['rank'],r['layers'],r['stage_ms_per_tick'],r['exchange_us_per_tick'],r['hop_us_one_row']) for r in d['pipeline_breakdown']['by_rank']])"
BENCH_SHARED_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 > $O/n2.out 2> $O/n2.err; echo "n2 (torchrun) rc=$? lines=$(wc -l < $O/n2.out)"; python -c "import json;d=json.load(open('$O/n2.out'));print(d['value'],d['parity'])"
