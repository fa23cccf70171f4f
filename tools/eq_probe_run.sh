#!/bin/bash
# round 5: k_stream_eq (fp32, one column tile, sixteen equal waves, folded norm) against k_stream_mm2 on the four 7B launches
O=gpurun_out/eq_probe.txt; : > $O
P=tools/stream_mm_check
echo "== checked: 352 x 1024 n=13, plain and folded norm" >> $O
timeout 120 $P 352 1024 13 64 6 >> $O 2>&1
STREAM_EQ_NORM=1 timeout 120 $P 352 1024 13 64 6 >> $O 2>&1
echo "== checked: 4096 x 4096 n=16 folded norm (wo-like)" >> $O
STREAM_EQ_NORM=1 timeout 300 $P 4096 4096 16 64 6 2>&1 | head -8 >> $O
export STREAM_CHECK_SKIP=1
for n in 9 16; do
  for sh in "22016 4096" "12288 4096" "4096 4096" "4096 11008"; do
    echo "== M K = $sh, n=$n: k_stream_eq (2/3/4/5 images), k_stream_mm2" >> $O
    for img in 2 3 4 5; do STREAM_DMA_IMAGES=$img timeout 120 $P $sh $n 64 6 2>&1 | grep "us per launch" >> $O; done
    STREAM_EQ_NORM=1 timeout 120 $P $sh $n 64 6 2>&1 | grep "us per launch" >> $O
    timeout 120 $P $sh $n 128 2 2>&1 | grep "us per launch" >> $O
  done
done
cat $O
