#!/bin/bash
# K-split of the single-tile stream launches (wo, w2): kernel time alone and with the reduce pass, S = 1, 2, 4 at 16 and 32 rows.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s17; mkdir -p $O
{ for n in 16 32; do
    for shape in "4096 4096" "4096 11008"; do
      for s in 1 2 4 8; do
        echo "== M K = $shape, N = $n, S = $s (KC 128)"; timeout 120 ./tools/stream_mm_check $shape $n 128 2 $s | head -8
      done
      echo "== M K = $shape, N = $n, S = 1 (KC 256)"; timeout 120 ./tools/stream_mm_check $shape $n 256 2 1 | head -6
    done
  done; } > $O/ksplit.txt 2>&1
grep -E "^==|us per|max abs" $O/ksplit.txt
