#!/bin/bash
OUT=gpurun_out/${1:-s15}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for N in ${NS:-16}; do
  rm -rf $OUT/prof_q8_n$N
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_q8_n$N -o n$N -- python $R/tools/bench_ttft.py --ns $N --reps 4 --int8 > $R/$OUT/prof_q8_n$N.log 2>&1 )
  db=$(find $OUT/prof_q8_n$N -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db 5 > $OUT/q8_n${N}_kernel_trace.txt 2>&1
  head -20 $OUT/q8_n${N}_kernel_trace.txt
done
find $OUT -name "*.db" -delete
