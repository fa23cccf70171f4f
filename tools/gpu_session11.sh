#!/bin/bash
# stream GEMM for 9..64 rows: parity tests, TTFT table with it on / off, kernel trace at N = 16
OUT=gpurun_out/${1:-s11}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -x -q -k "prefill or odd_shapes or chunked or reproducible or config3" > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log
timeout 300 python tools/bench_ttft.py --ns 8,9,16,24,32,48,64,128 > $OUT/ttft_on.json 2>> $OUT/ttft.err; cat $OUT/ttft_on.json
LLAMAHIP_STREAM_MM=0 timeout 300 python tools/bench_ttft.py --ns 9,16,32,64 > $OUT/ttft_off.json 2>> $OUT/ttft.err; cat $OUT/ttft_off.json
for N in ${NS:-16 64}; do
  rm -rf $OUT/prof_n$N
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_n$N -o n$N -- python $R/tools/bench_ttft.py --ns $N --reps 4 > $R/$OUT/prof_n$N.log 2>&1 )
  db=$(find $OUT/prof_n$N -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db 5 > $OUT/n${N}_kernel_trace.txt 2>&1
  head -16 $OUT/n${N}_kernel_trace.txt
done
find $OUT -name "*.db" -delete
