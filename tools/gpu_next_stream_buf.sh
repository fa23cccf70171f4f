#!/bin/bash
# First GPU call of the next round: the stream kernel's loader on buffer loads (and with a short sleep behind the chunk barrier), and its
# MFMA waves paced with s_nop, against the shipped kernel - standalone, CHECKED against the float64 host product, on the 7B launches at 16 / 32 / 48 / 64 rows.
# Build first (CPU side, ~6 min): bash tools/build_probes.sh
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/next_stream_buf; mkdir -p $O
{ for shape in "22016 4096" "12288 4096"; do for cfg in "16 128" "32 128" "48 128" "64 64"; do for b in stream_mm_check stream_mm_check_buf stream_mm_check_buf_sleep6 stream_mm_check_pace6 stream_mm_check_pace8 stream_mm_check_pace9 stream_mm_check_pace10 stream_mm_check_pace11 stream_mm_check_buf_pace9; do
    echo "== M K = $shape, N KC = $cfg, $b"; STREAM_CHECK_SKIP=1 timeout 30 ./tools/$b $shape $cfg 2 | grep -E "us per launch|clocks per chunk|MFMA wave"
  done; done; done
  echo "== what a wave can issue next to an MFMA wave of its SIMD"; timeout 60 ./tools/valu_mfma_probe 2000 4000 5 13
  for a in "4096 4096 16 128 2 1" "4096 4096 48 128 2 2" "4096 11008 32 128 2 2" "1024 2816 64 64 2 1" "256 1024 33 128 2 1"; do
    echo "== checked: stream_mm_check_buf $a"; timeout 60 ./tools/stream_mm_check_buf $a | grep -E "us per|max abs|wrong"
  done; } > $O/buf.txt 2>&1
grep -E "^==|us per launch|max abs|clocks|load|add|write" $O/buf.txt
