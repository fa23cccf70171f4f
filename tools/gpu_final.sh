#!/bin/bash
# Round-end evidence run: full GPU test-suite, the default bench line, rocprofv3 kernel traces and PMC passes (each counter set in its
# own run, --kernel-trace only), TTFT tables.  Output under gpurun_out/$1/; summaries are copied to profiles/ by hand afterwards.
OUT=gpurun_out/${1:-final}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
fi
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
Q="--no-cpu-baseline --no-prefill"
prof() { # name, command...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$name -o $name -- "$@" > $R/$OUT/prof_$name.log 2>&1 )
  local db=$(find $OUT/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db ${NSTEPS:-16} > $OUT/${name}_kernel_trace.txt 2>&1
}
pmc() { # name, counters, command...
  local name=$1; local ctr=$2; shift; shift
  rm -rf $OUT/pmc_$name
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctr -d $R/$OUT/pmc_$name -o $name -- "$@" > $R/$OUT/pmc_$name.log 2>&1 )
}
prof f32 python $R/bench.py $Q
prof q8 python $R/bench.py --int8 $Q
prof ttft8 python $R/tools/bench_ttft.py --ns 8 --reps 4
prof p13 python $R/tools/bench_prefill.py --shape 13B --n 1024
pmc fetch FETCH_SIZE python $R/bench.py $Q
db=$(find $OUT/pmc_fetch -name "*.db" | head -1); [ -n "$db" ] && python tools/pmc_summary.py $db $OUT/pmc_traffic.json > $OUT/pmc_fetch_size.txt 2>&1
pmc q8valu "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" python $R/bench.py --int8 $Q
db=$(find $OUT/pmc_q8valu -name "*.db" | head -1); [ -n "$db" ] && python tools/pmc_dump.py $db k_gemv_q8 > $OUT/q8_pmc_valu.txt 2>&1
pmc q8wait "SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" python $R/bench.py --int8 $Q
db=$(find $OUT/pmc_q8wait -name "*.db" | head -1); [ -n "$db" ] && python tools/pmc_dump.py $db k_gemv_q8 > $OUT/q8_pmc_wait.txt 2>&1
pmc p13mfma "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" python $R/tools/bench_prefill.py --shape 13B --n 1024 --layers 8
db=$(find $OUT/pmc_p13mfma -name "*.db" | head -1); [ -n "$db" ] && python tools/pmc_dump.py $db > $OUT/p13_pmc_mfma.txt 2>&1
timeout 300 python tools/bench_ttft.py --ns 1,2,4,8,9,16,32,64,128 > $OUT/ttft_f32.json 2>> $OUT/ttft.err
timeout 300 python tools/bench_ttft.py --ns 1,2,4,8,9,16,32,64,128 --int8 > $OUT/ttft_q8.json 2>> $OUT/ttft.err
LLAMAHIP_SKINNY=0 timeout 300 python tools/bench_ttft.py --ns 2,4,8 > $OUT/ttft_f32_skinny0.json 2>> $OUT/ttft.err
timeout 300 python tools/bench_longctx.py > $OUT/longctx.txt 2>&1
find $OUT -name "*.db" -size +20M -delete
python - "$OUT" <<'PY'
import json,sys
o=sys.argv[1]
try:
    d=json.loads(open(o+"/bench_default.json").read().strip().splitlines()[-1])
    print("default", d["value"], d["roofline_token"]["frac_of_hbm_roofline"], d["roofline"]["frac"], d["parity"]["token_ids_match"], d["parity"]["steps_compared"], d.get("int8_decode"), d.get("prompt_8_tokens"), d.get("prefill_13b",{}).get("frac_of_fp32_mfma_peak_157.3"))
except Exception as e: print("ERR", e)
PY
tail -12 $OUT/f32_kernel_trace.txt; cat $OUT/pmc_fetch_size.txt; cat $OUT/ttft_f32.json $OUT/ttft_q8.json
