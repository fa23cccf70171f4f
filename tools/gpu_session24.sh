#!/bin/bash
# Probe 16 (loader addresses from SGPR bases, no vector ALU in front of the loads) vs the product kernel: w1|w3 at 16 / 48 rows, one checked run.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s24; mkdir -p $O
{ for cfg in "16 128" "48 128"; do for b in stream_mm_check stream_mm_check_p16; do
    echo "== w1|w3 22016 x 4096, N KC = $cfg, $b"; STREAM_CHECK_SKIP=1 timeout 15 ./tools/$b 22016 4096 $cfg 2 | grep -E "us per launch|clocks per chunk|MFMA wave"
  done; done
  echo "== checked: 4096 x 4096, N 16, stream_mm_check_p16"; timeout 20 ./tools/stream_mm_check_p16 4096 4096 16 128 2 | grep -E "us per|max abs"; } > $O/p16.txt 2>&1
cat $O/p16.txt
