#!/bin/bash
# round 4, GPU call 29: k_stream_q8 with the scale multiplies of the conversion packed (v_pk_mul_f32, -DQ8_CVT=3) against the product's form, standalone
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}" || exit 1
O=gpurun_out/r4q8cvt3; mkdir -p $O
{
echo "### checked against the CPU sum (K = 512), packed build"
for shape in "22016 512" "4096 1024"; do for n in 8 16 48; do echo "--- $shape n=$n"; timeout 120 tools/stream_mm_check_q8cvt3 $shape $n 256 5 2>&1 | grep -E "k_stream_q8|max abs|do not fit|first wrong|HIP error"; done; done
export STREAM_CHECK_SKIP=1
for round in 1 2; do for shape in "22016 4096" "12288 4096" "4096 11008" "4096 4096"; do for n in 8 16 32; do
  echo "--- round $round shape $shape n=$n"
  for b in "" _q8cvt3; do echo -n "build '$b': "; timeout 60 tools/stream_mm_check$b $shape $n 256 5 2>&1 | grep -E "us per launch|do not fit"; done
done; done; done
} > $O/q8_cvt3.log 2>&1
cat $O/q8_cvt3.log | tail -60
