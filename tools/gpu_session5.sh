#!/bin/bash
OUT=gpurun_out/${1:-s5}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
R=$PWD
prof() { # name, command...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$name -o $name -- "$@" > $OUT/prof_$name.log 2>&1
  local db=$(find $OUT/prof_$name -name "*.db" | head -1)
  if [ -n "$db" ]; then python tools/prof_summary.py $db ${NSTEPS:-16} > $OUT/${name}_kernel_trace.txt 2>&1; fi
  find $OUT/prof_$name -name "*.db" -size +30M -delete
}
prof ttft8 python tools/bench_ttft.py --ns 8 --reps 4
prof q8 python bench.py --int8 --no-cpu-baseline --no-prefill
prof f32 python bench.py --no-cpu-baseline --no-prefill
head -40 $OUT/ttft8_kernel_trace.txt
tail -22 $OUT/q8_kernel_trace.txt
tail -14 $OUT/f32_kernel_trace.txt
ls -la $OUT
