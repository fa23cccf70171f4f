#!/bin/bash
O=gpurun_out/q8b_probe_threads.txt; : > $O
export Q8B_ONLY=1
echo "== checked, 512 threads: w1|w3 n=8" >> $O; timeout 300 tools/q8b_probe_th512 11008 4096 8 256 2 1 >> $O 2>&1
export Q8B_SKIP_CHECK=1
for n in 8 16 32; do
  for P in tools/q8b_probe tools/q8b_probe_th512; do
  echo "== $P w1|w3 n=$n" >> $O; timeout 120 $P 11008 4096 $n 256 2 1 >> $O 2>&1
  echo "== $P wq|wk|wv n=$n kc256" >> $O; timeout 120 $P 4096 4096 $n 256 3 0 >> $O 2>&1
  echo "== $P wo ksplit 4 n=$n" >> $O; timeout 120 $P 4096 4096 $n 256 1 0 4 >> $O 2>&1
  echo "== $P w2 ksplit 4 n=$n" >> $O; timeout 120 $P 4096 11008 $n 256 1 0 4 >> $O 2>&1
  done
done
grep -v "^split3\|^M \|wave 15\|wave  7" $O
