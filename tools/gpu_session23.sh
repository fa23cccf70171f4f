#!/bin/bash
# Loader waves at s_setprio 3 (a StreamArgs::prio switch that existed for this run only; measured without effect and removed): standalone A/B, one checked run, TTFT A/B, prefill parity with the switch on.
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/s23; mkdir -p $O
{ for cfg in "16 128" "32 128" "48 128" "64 64"; do for pr in 0 1; do
    echo "== w1|w3 22016 x 4096, N KC = $cfg, prio $pr"; STREAM_CHECK_SKIP=1 timeout 20 ./tools/stream_mm_check 22016 4096 $cfg 2 1 $pr | grep -E "us per launch|clocks per chunk|MFMA wave"
  done; done
  echo "== checked run: 4096 x 4096, N 48, K-split 2, prio 1"; timeout 30 ./tools/stream_mm_check 4096 4096 48 128 2 2 1 | grep -E "us per|max abs"; } > $O/prio.txt 2>&1
grep -E "^==|us per|max abs" $O/prio.txt
{ echo "prio 1"; LLAMAHIP_STREAM_PRIO=1 timeout 30 python tools/bench_ttft.py --ns 8,16,32,48,64 --reps 2 2>/dev/null | tail -1
  echo "prio 0"; timeout 30 python tools/bench_ttft.py --ns 8,16,32,48,64 --reps 2 2>/dev/null | tail -1; } | tee $O/ttft_prio.txt
LLAMAHIP_STREAM_PRIO=1 timeout 40 python -m pytest tests/test_gpu_llama.py -m gpu -x -q -k "prefill_mfma or short_prompts" 2>&1 | tail -2 | tee $O/pytest_prio.txt
