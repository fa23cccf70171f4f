#!/bin/bash
# round 4, GPU call 17: shader clock under fp32 MFMA load - the bare MFMA loop with random operands, and inside k_gemm_glds on the 13B prefill shapes (random data)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4clock4; mkdir -p $O
timeout 60 tools/mfma_clock_probe 200000 1 > $O/mfma_clock.txt 2>&1; echo "probe rc=$?"; cat $O/mfma_clock.txt
timeout 120 tools/gemm_probe_clock > $O/gemm_clock.txt 2>&1; echo "gemm rc=$?"; grep -A1 'shader clock' $O/gemm_clock.txt | grep -v '^--'
