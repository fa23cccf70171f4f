#!/bin/bash
# Round-2 (second half) evidence run: full GPU test-suite, the default bench line, rocprofv3 kernel traces (decode fp32 / int8, short
# prompts on the stream kernel, 13B prefill), the HBM-traffic PMC pass, TTFT tables with the stream kernel on and off, the standalone
# checkers / probes.  Output under gpurun_out/$1/; summaries are copied to profiles/ by hand afterwards.
OUT=gpurun_out/${1:-final2}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
python -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1800 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; grep -n "passed\|failed" $OUT/pytest.log | tail -2
fi
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
Q="--no-cpu-baseline --no-prefill"
prof() { # name, nsteps-for-summary, command...
  local name=$1; local ns=$2; shift; shift
  rm -rf $OUT/prof_$name
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/prof_$name -o $name -- "$@" > $R/$OUT/prof_$name.log 2>&1 )
  local db=$(find $OUT/prof_$name -name "*.db" | head -1)
  [ -n "$db" ] && python tools/prof_summary.py $db $ns > $OUT/${name}_kernel_trace.txt 2>&1
}
prof f32 16 python $R/bench.py $Q
prof q8 16 python $R/bench.py --int8 $Q
prof ttft8 5 python $R/tools/bench_ttft.py --ns 8 --reps 4
prof ttft16 5 python $R/tools/bench_ttft.py --ns 16 --reps 4
prof ttft32 5 python $R/tools/bench_ttft.py --ns 32 --reps 4
prof ttft16q8 5 python $R/tools/bench_ttft.py --ns 16 --reps 4 --int8
prof p13 16 python $R/tools/bench_prefill.py --shape 13B --n 1024
rm -rf $OUT/pmc_fetch
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$OUT/pmc_fetch -o fetch -- python $R/bench.py $Q > $R/$OUT/pmc_fetch.log 2>&1 )
db=$(find $OUT/pmc_fetch -name "*.db" | head -1); [ -n "$db" ] && python tools/pmc_summary.py $db $OUT/pmc_traffic.json > $OUT/pmc_fetch_size.txt 2>&1
timeout 300 python tools/bench_ttft.py --ns 1,2,4,8,9,16,24,32,33,64,128 > $OUT/ttft_f32.json 2>> $OUT/ttft.err
LLAMAHIP_STREAM_MM=0 timeout 300 python tools/bench_ttft.py --ns 2,4,8,9,16,24,32 > $OUT/ttft_f32_stream0.json 2>> $OUT/ttft.err
LLAMAHIP_STREAM_MM=1 timeout 300 python tools/bench_ttft.py --ns 9,16,32 > $OUT/ttft_f32_stream_v1.json 2>> $OUT/ttft.err
LLAMAHIP_STREAM_FUSED=0 timeout 300 python tools/bench_ttft.py --ns 8,16,32 > $OUT/ttft_f32_unfused.json 2>> $OUT/ttft.err
timeout 300 python tools/bench_ttft.py --ns 1,2,3,4,8,9,16,32,64,128 --int8 > $OUT/ttft_q8.json 2>> $OUT/ttft.err
LLAMAHIP_STREAM_MM=0 timeout 300 python tools/bench_ttft.py --ns 4,8,16,32 --int8 > $OUT/ttft_q8_stream0.json 2>> $OUT/ttft.err
{ for a in "22016 4096 16 128" "12288 4096 16 128" "4096 4096 16 128" "4096 11008 16 256" "22016 4096 32 128" "12288 4096 32 128"; do echo "== stream_mm_check $a (first variant, then wave-specialised)"; ./tools/stream_mm_check $a 0 | head -4 | tail -3; ./tools/stream_mm_check $a 2 | head -5 | tail -4; done;
  for a in "4096 4096 16 512" "22016 4096 16 256"; do echo "== longer chunks: stream_mm_check $a"; ./tools/stream_mm_check $a 0 | head -4 | tail -3; done;
  for a in "4096 4096 16 128" "22016 4096 16 128"; do echo "== chunk-major weight copy: stream_mm_check $a 1"; ./tools/stream_mm_check $a 1 | head -4 | tail -3; done; } > $OUT/stream_mm_check.txt 2>&1
timeout 120 ./tools/resident_probe > $OUT/resident_trace.txt 2>&1
PROBE_SKIP_PERSIST=1 timeout 120 ./tools/persist_probe > $OUT/persist_probe_attn_wo.txt 2>&1
timeout 300 python tools/bench_longctx.py > $OUT/longctx.txt 2>&1
find $OUT -name "*.db" -delete
python - "$OUT" <<'PY'
import json,sys
o=sys.argv[1]
try:
    d=json.loads(open(o+"/bench_default.json").read().strip().splitlines()[-1])
    print("default", d["value"], d["roofline_token"]["frac_of_hbm_roofline"], d["roofline"]["frac"], d["parity"]["token_ids_match"], d["parity"]["steps_compared"], d.get("int8_decode"), d.get("prompt_8_tokens"), d.get("prefill_13b",{}).get("frac_of_fp32_mfma_peak_157.3"))
except Exception as e: print("ERR", e)
PY
cat $OUT/ttft_f32.json $OUT/ttft_f32_stream0.json $OUT/ttft_q8.json $OUT/ttft_q8_stream0.json
