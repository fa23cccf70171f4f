#!/bin/bash
# same-box A/B of two library builds (lib = new, lib_old = baseline) on pod ticks and short prompts, fp32 and block-int8, alternating processes
out=${1:-gpurun_out/r6abp}; mkdir -p $out
swap() { (cd llama.go_amd && mv lib lib_tmp && mv lib_old lib && mv lib_tmp lib_old); }
for rep in 1 2; do
  for which in new old; do
    echo "== $which rep $rep" >> $out/ab.txt
    python tools/bench_pods.py --pods 2,4,8,16,32,64 --steps 32 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print(d['shape'], ' '.join(f\"{k}:{v['ms_per_step']:.3f}\" for k, v in d['by_pods'].items()))
" >> $out/ab.txt
    python tools/bench_pods.py --int8 --pods 2,4,8,16,32,64 --steps 32 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l)
        print(d['shape'], ' '.join(f\"{k}:{v['ms_per_step']:.3f}\" for k, v in d['by_pods'].items()))
" >> $out/ab.txt
    swap
  done
done
cat $out/ab.txt
