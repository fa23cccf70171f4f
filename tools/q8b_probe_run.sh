#!/bin/bash
O=gpurun_out/q8b_probe.txt; : > $O
P=tools/q8b_probe
export Q8B_ONLY=1
echo "== w1|w3 silu*mul, n=8 (checked)" >> $O; timeout 300 $P 11008 4096 8 256 2 1 >> $O 2>&1
echo "== wq|wk|wv, n=32 kc128 (checked)" >> $O; timeout 300 $P 4096 4096 32 128 3 0 >> $O 2>&1
echo "== wq|wk|wv, n=5 kc512 (checked)" >> $O; timeout 300 $P 4096 4096 5 512 3 0 >> $O 2>&1
echo "== w2 ksplit 4, n=20 (checked)" >> $O; timeout 300 $P 4096 11008 20 256 1 0 4 >> $O 2>&1
export Q8B_SKIP_CHECK=1
for n in 8 16 32 64; do
  echo "== w1|w3 n=$n" >> $O; timeout 120 $P 11008 4096 $n 256 2 1 >> $O 2>&1
  echo "== wq|wk|wv n=$n" >> $O; timeout 120 $P 4096 4096 $n 256 3 0 >> $O 2>&1
  echo "== wq|wk|wv n=$n kc512" >> $O; timeout 120 $P 4096 4096 $n 512 3 0 >> $O 2>&1
  echo "== wo n=$n kc512" >> $O; timeout 120 $P 4096 4096 $n 512 1 0 >> $O 2>&1
  echo "== wo ksplit 2 n=$n" >> $O; timeout 120 $P 4096 4096 $n 256 1 0 2 >> $O 2>&1
  echo "== wo ksplit 4 n=$n" >> $O; timeout 120 $P 4096 4096 $n 256 1 0 4 >> $O 2>&1
  echo "== w2 ksplit 2 n=$n" >> $O; timeout 120 $P 4096 11008 $n 256 1 0 2 >> $O 2>&1
  echo "== w2 ksplit 4 n=$n" >> $O; timeout 120 $P 4096 11008 $n 256 1 0 4 >> $O 2>&1
done
echo "== lm_head n=8" >> $O; timeout 120 $P 32000 4096 8 256 1 0 >> $O 2>&1
grep -v "^split3\|^M \|wave 15" $O
