#!/bin/bash
# round 4, GPU call 7: where k_stream_q8's time goes (timing-only ablation builds: -DQ8_ABL=1 no conversion, 2 no MFMAs, 3 neither)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4c7; mkdir -p $O
export STREAM_CHECK_SKIP=1
{
for round in 1 2; do for shape in "22016 4096" "12288 4096" "4096 11008"; do for n in 16 32; do
  echo "--- round $round shape $shape n=$n"
  for b in "" _q8abl1 _q8abl2 _q8abl3; do for kc in 256 128; do echo -n "build '$b' KC=$kc: "; timeout 60 tools/stream_mm_check$b $shape $n $kc 5 2>&1 | grep -E "us per launch|do not fit"; done; done
done; done; done
} > $O/q8_ablation.log 2>&1
tail -20 $O/q8_ablation.log
unset STREAM_CHECK_SKIP
timeout 600 python -m pytest tests/test_gpu_sample.py tests/test_gpu_pipeline.py -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $O/tests.log | tail -8
