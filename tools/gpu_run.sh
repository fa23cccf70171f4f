#!/bin/bash
# tools/gpu_run.sh — the ONE parametrised runner for GPU sessions (replaces the per-session scripts of rounds 1-2).
#   gpurun --timeout 1500 -- 'bash tools/gpu_run.sh <name> step [step ...]'
# Every step writes gpurun_out/<name>/<step>.log (+ what the step itself produces); a step that fails does not stop the next one.
#   tests[:<pytest -k expr>]  pytest -m gpu (optionally filtered)             bench[:<args>]   python bench.py <args> -> bench.json
#   trace:<tag>:<cmd>         rocprofv3 --kernel-trace --stats of <cmd>        pmc:<tag>:<ctrs>:<cmd>  rocprofv3 --pmc <ctrs> of <cmd>
#   sh:<tag>:<cmd>            any shell command
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
name=$1; shift
O=gpurun_out/$name; mkdir -p "$O"
export TMPDIR=/tmp
for step in "$@"; do
  kind=${step%%:*}; rest=${step#*:}; [ "$rest" = "$step" ] && rest=""
  t0=$(date +%s)
  case $kind in
    tests) if [ -n "$rest" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -rs -k "$rest" > "$O/tests.log" 2>&1; else timeout 1500 python -m pytest tests -m gpu -x -q -rs > "$O/tests.log" 2>&1; fi
           echo "rc=$?" >> "$O/tests.log"; grep -E "^(FAILED|ERROR)|passed|failed|^rc=" "$O/tests.log" | tail -8 ;;   # (a tail would show RCCL's exit banner, not the verdict)
    bench) timeout 900 python bench.py $rest > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"; tail -c 1500 "$O/bench.json" ;;
    trace) tag=${rest%%:*}; cmd=${rest#*:}
           (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$OLDPWD/$O/prof_$tag" -o "$tag" -- bash -c "cd $OLDPWD && $cmd") > "$O/trace_$tag.log" 2>&1; echo "trace $tag rc=$?"
           python tools/prof_summary.py "$(find "$O/prof_$tag" -name '*.db' | head -1)" > "$O/trace_$tag.txt" 2>&1; head -30 "$O/trace_$tag.txt"
           find "$O/prof_$tag" -name "*_kernel_trace.csv" -size +20M -delete ;;
    pmc)   tag=${rest%%:*}; r2=${rest#*:}; ctrs=${r2%%:*}; cmd=${r2#*:}
           (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $ctrs -d "$OLDPWD/$O/pmc_$tag" -o "$tag" -- bash -c "cd $OLDPWD && $cmd") > "$O/pmc_$tag.log" 2>&1; echo "pmc $tag rc=$?"
           python tools/pmc_summary.py "$(find "$O/pmc_$tag" -name '*.db' | head -1)" "$O/pmc_$tag.json" > "$O/pmc_$tag.txt" 2>&1; head -30 "$O/pmc_$tag.txt"
           find "$O/pmc_$tag" -name "*.csv" -size +20M -delete ;;
    sh)    tag=${rest%%:*}; cmd=${rest#*:}; timeout 900 bash -c "$cmd" > "$O/$tag.log" 2>&1; echo "$tag rc=$?"; tail -25 "$O/$tag.log" ;;
    *) echo "unknown step $step" ;;
  esac
  echo "== step $kind took $(( $(date +%s) - t0 )) s"
done
