#!/bin/bash
# Full GPU suite + default bench + TTFT table on the current tree.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s16; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 300 python tools/bench_ttft.py --ns 2,4,8,16,24,32,33,48,64,128 > $O/ttft.json 2>&1
cat $O/rc.txt; tail -3 $O/pytest.log; cat $O/ttft.json
