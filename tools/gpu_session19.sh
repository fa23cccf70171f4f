#!/bin/bash
# 33..48 rows on the stream kernel (three column tiles): parity subset, then TTFT with the limit at 32 / 48 rows and 2- / 4-way K-split.
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s19; mkdir -p $O
timeout 700 python -m pytest tests/test_gpu_llama.py -m gpu -x -q -k "prefill or odd_shapes or chunked or reproducible or int8 or 7b_shape or stage" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
{ echo "limit 32 rows (tile GEMM from 33)"; LLAMAHIP_STREAM_MAX_ROWS=32 timeout 300 python tools/bench_ttft.py --ns 33,40,48,49 2>/dev/null | tail -1
  echo "limit 48, K-split 4 (default)"; timeout 300 python tools/bench_ttft.py --ns 32,33,40,48,49 2>/dev/null | tail -1
  echo "limit 48, K-split 2"; LLAMAHIP_STREAM_KSPLIT_S=2 timeout 300 python tools/bench_ttft.py --ns 33,40,48 2>/dev/null | tail -1
  echo "limit 48, no K-split"; LLAMAHIP_STREAM_KSPLIT=0 timeout 300 python tools/bench_ttft.py --ns 33,40,48 2>/dev/null | tail -1
  echo "int8 limit 32"; LLAMAHIP_STREAM_MAX_ROWS=32 timeout 300 python tools/bench_ttft.py --ns 33,48 --int8 2>/dev/null | tail -1
  echo "int8 limit 48"; timeout 300 python tools/bench_ttft.py --ns 33,48 --int8 2>/dev/null | tail -1; } | tee $O/ttft_48.txt
{ for shape in "22016 4096" "12288 4096"; do echo "== M K = $shape, N = 48"; timeout 120 ./tools/stream_mm_check $shape 48 128 2 1 | head -6; done
  for shape in "4096 4096" "4096 11008"; do for s in 1 2 4; do echo "== M K = $shape, N = 48, S = $s"; timeout 120 ./tools/stream_mm_check $shape 48 128 2 $s | head -7; done; done; } > $O/stream48.txt 2>&1
grep -E "^==|us per|max abs" $O/stream48.txt
