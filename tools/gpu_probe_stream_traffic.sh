#!/bin/bash
# What blocks the loader waves of k_stream_mm2 at 33..64 rows?  The same launch with one traffic class removed at a time (binaries from
# tools/build_probes.sh): p1 X re-read from L1, p2 X non-temporal, p4 W temporal, p8 W from cache, p9 neither stream (LDS + MFMA only).
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/probe_stream; mkdir -p $O
export STREAM_CHECK_SKIP=1
IFS=';' read -ra SH <<< "${PROBE_SHAPES:-22016 4096;12288 4096}"
{ for shape in "${SH[@]}"; do
    for cfg in "16 128" "32 128" "48 128" "64 64"; do
      for p in "" _p1 _p2 _p4 _p8 _p9; do
        echo "== M K = $shape, N KC = $cfg, probe ${p:-none}"; timeout 60 ./tools/stream_mm_check$p $shape $cfg 2 1 | grep -E "us per launch|clocks per chunk|MFMA wave"
      done
    done
  done; } > $O/traffic_probe.txt 2>&1
grep -E "^==|us per launch" $O/traffic_probe.txt
