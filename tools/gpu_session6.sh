#!/bin/bash
OUT=gpurun_out/${1:-s6}
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_llama.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
LLAMAHIP_SKINNY=1 timeout 300 python tools/bench_ttft.py --ns 1,2,4,8,9 > $OUT/ttft_skinny1.json 2>>$OUT/ttft.err
cat $OUT/ttft_skinny1.json
for fl in 0 1; do
LLAMAHIP_FLASH=$fl timeout 300 python tools/bench_prefill.py --shape 13B --n 1024 > $OUT/prefill13b_flash$fl.json 2>>$OUT/prefill.err; cat $OUT/prefill13b_flash$fl.json
LLAMAHIP_FLASH=$fl timeout 300 python tools/bench_prefill.py --shape 7B --n 512 > $OUT/prefill7b_flash$fl.json 2>>$OUT/prefill.err; cat $OUT/prefill7b_flash$fl.json
done
rm -rf $OUT/prof_p13; cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_p13 -o p13 -- python $OLDPWD/tools/bench_prefill.py --shape 13B --n 1024 --layers 8 > $OLDPWD/$OUT/prof_p13.log 2>&1; cd $OLDPWD
db=$(find $OUT/prof_p13 -name "*.db" | head -1); [ -n "$db" ] && python tools/prof_summary.py $db > $OUT/p13_kernel_trace.txt 2>&1; head -24 $OUT/p13_kernel_trace.txt
find $OUT -name "*.db" -size +30M -delete
