#!/bin/bash
# 49..64 rows on the stream kernel (four column tiles, 64-column chunks): full GPU suite, then the TTFT A/B against the tile GEMM on the same box.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s22; mkdir -p $O
timeout 330 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" > $O/rc.txt
{ echo "default (stream kernel up to 64 rows)"; timeout 40 python tools/bench_ttft.py --ns 48,49,56,64,65 2>/dev/null | tail -1
  echo "LLAMAHIP_STREAM_MAX_ROWS=48 (tile GEMM from 49 rows)"; LLAMAHIP_STREAM_MAX_ROWS=48 timeout 40 python tools/bench_ttft.py --ns 49,56,64 2>/dev/null | tail -1; } > $O/ttft_64.txt
cat $O/rc.txt; grep -n "passed\|failed" $O/pytest.log | tail -2; cat $O/ttft_64.txt
