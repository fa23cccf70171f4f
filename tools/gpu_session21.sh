#!/bin/bash
# Final verification of the round: full GPU suite, default bench line, TTFT table.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s21; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
timeout 200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/rc.txt
timeout 60 python tools/bench_ttft.py --ns 2,8,16,24,32,33,40,48,49,64 > $O/ttft.json 2>/dev/null
cat $O/rc.txt; grep -n "passed\|failed" $O/pytest.log | tail -2; tail -1 $O/ttft.json
