#!/bin/bash
O=gpurun_out/q8b_probe.txt; : > $O
P=tools/q8b_probe
export Q8B_ONLY=1
echo "== w1|w3 silu*mul, n=8 (checked)" >> $O; timeout 300 $P 11008 4096 8 256 2 1 >> $O 2>&1
echo "== wq|wk|wv, n=32 kc128 (checked)" >> $O; timeout 300 $P 4096 4096 32 128 3 0 >> $O 2>&1
echo "== wq|wk|wv, n=5 kc512 (checked)" >> $O; timeout 300 $P 4096 4096 5 512 3 0 >> $O 2>&1
echo "== w2 ksplit 4, n=20 (checked)" >> $O; timeout 300 $P 4096 11008 20 256 1 0 4 >> $O 2>&1
export Q8B_SKIP_CHECK=1
for n in 8 16 32; do
  for img in 2 3 0; do export Q8B_IMAGES=$img
  echo "== w1|w3 n=$n images $img" >> $O; timeout 120 $P 11008 4096 $n 256 2 1 >> $O 2>&1
  echo "== wq|wk|wv n=$n kc256 images $img" >> $O; timeout 120 $P 4096 4096 $n 256 3 0 >> $O 2>&1
  echo "== wo ksplit 4 n=$n images $img" >> $O; timeout 120 $P 4096 4096 $n 256 1 0 4 >> $O 2>&1
  echo "== w2 ksplit 4 n=$n images $img" >> $O; timeout 120 $P 4096 11008 $n 256 1 0 4 >> $O 2>&1
  done
  unset Q8B_IMAGES
  echo "== wq|wk|wv n=$n kc512" >> $O; timeout 120 $P 4096 4096 $n 512 3 0 >> $O 2>&1
done
grep -v "^split3\|^M \|wave 15\|wave  0" $O
