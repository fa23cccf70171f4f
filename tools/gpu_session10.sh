#!/bin/bash
# kernel traces of one Eval at N = 16 and N = 64 (7B fp32)
OUT=gpurun_out/${1:-s10}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
for N in ${NS:-16 64}; do
  rm -rf $OUT/prof_n$N
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_n$N -o n$N -- python $R/tools/bench_ttft.py --ns $N --reps 4 > $R/$OUT/prof_n$N.log 2>&1 )
  db=$(find $OUT/prof_n$N -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db 5 > $OUT/n${N}_kernel_trace.txt 2>&1
  tail -3 $OUT/prof_n$N.log
  head -60 $OUT/n${N}_kernel_trace.txt
done
find $OUT -name "*.db" -size +20M -delete
