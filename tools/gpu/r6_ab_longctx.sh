#!/bin/bash
# same-box A/B of two library builds on decode behind long contexts (split-T attention): lib (new) against lib_old
out=${1:-gpurun_out/r6ablc}; mkdir -p $out
swap() { (cd llama.go_amd && mv lib lib_tmp && mv lib_old lib && mv lib_tmp lib_old); }
for rep in 1 2; do
  for which in new old; do
    echo "== $which rep $rep" >> $out/ab.txt
    python tools/bench_longctx.py --past 120 500 1000 2000 --pods 8 2>&1 | grep -E "^\[" | python -c "
import sys, json
for l in sys.stdin:
    for e in json.loads(l):
        print(e)
" >> $out/ab.txt
    swap
  done
done
cat $out/ab.txt
