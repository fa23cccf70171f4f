#!/bin/bash
# round 6: k_stream_b9 before / after the in-place fragment schedule (tools/b9s_probe_old = the header of commit "lm_head halves", tools/b9s_probe_new = the tree), 8 products, the 7B launches
cd "$(dirname "$0")/.." || exit 1
OUT=../gpurun_out/r6_b9s_ab.txt
: > $OUT
run() { echo "== $*" >> $OUT; for b in old new old new; do echo "-- $b" >> $OUT; timeout 120 ./b9s_probe_$b "$@" 2>&1 | grep -v "^M \|split3" >> $OUT; done; }
run 352 1024 50 -
run 2048 512 64 - 3
run 4096 512 17 - 1 0 2
for n in 49 64; do
  run 11008 4096 $n - 2 1
  run 4096 4096 $n - 3
  run 4096 4096 $n - 1 0 4
  run 4096 11008 $n - 1 0 4
  run 16000 4096 $n -
done
for n in 32 48; do
  run 11008 4096 $n - 2 1
  run 4096 4096 $n - 3
done
cat $OUT
