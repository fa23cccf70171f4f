#!/bin/bash
OUT=gpurun_out/${1:-s8}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -x -q -k "greedy or odd or chunked or reference or golden or bitwise" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
LLAMAHIP_SKINNY=1 timeout 300 python tools/bench_ttft.py --ns 2,8 > $OUT/ttft_skinny1.json 2>>$OUT/ttft.err
cat $OUT/ttft_skinny1.json
R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_ttft8 -o ttft8 -- python $R/tools/bench_ttft.py --ns 8 --reps 4 > $R/$OUT/prof_ttft8.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/pmc_ttft8 -o pmc -- python $R/tools/bench_ttft.py --ns 8 --reps 2 > $R/$OUT/pmc_ttft8.log 2>&1
cd $R
db=$(find $OUT/prof_ttft8 -name "*.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py $db > $OUT/ttft8_kernel_trace.txt 2>&1; head -12 $OUT/ttft8_kernel_trace.txt
db=$(find $OUT/pmc_ttft8 -name "*.db" | head -1); [ -n "$db" ] && python tools/pmc_dump.py $db > $OUT/ttft8_pmc.txt 2>&1; head -30 $OUT/ttft8_pmc.txt
find $OUT -name "*.db" -size +30M -delete
