#!/bin/bash
# round 6: what the chip draws and clocks under (a) 64 pods on k_stream_b9, (b) 32 pods on k_stream_dma, (c) single-stream decode, (d) 13B prefill on k_gemm_b9 - rocm-smi READ-ONLY samples
cd "$(dirname "$0")/../.." || exit 1
OUT=gpurun_out/r6_power.txt
: > $OUT
sample() { for i in $(seq 1 $1); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|fclk" | tr '\n' ' ' | sed 's/  */ /g' >> $OUT; echo >> $OUT; sleep 0.5; done; }
echo "## idle" >> $OUT; sample 2
echo "## 64 pods (k_stream_b9): python tools/bench_pods.py --pods 64 --steps 1500" >> $OUT
python tools/bench_pods.py --pods 64 --steps 1500 > gpurun_out/r6_power_p64.log 2>&1 & P=$!; sleep 9; sample 8; wait $P; tail -1 gpurun_out/r6_power_p64.log | cut -c1-300 >> $OUT
echo "## 32 pods (k_stream_dma): --pods 32 --steps 2000" >> $OUT
python tools/bench_pods.py --pods 32 --steps 2000 > gpurun_out/r6_power_p32.log 2>&1 & P=$!; sleep 9; sample 8; wait $P; tail -1 gpurun_out/r6_power_p32.log | cut -c1-300 >> $OUT
echo "## 64 pods on k_stream_dma (LLAMAHIP_B9S_MIN=65)" >> $OUT
LLAMAHIP_B9S_MIN=65 python tools/bench_pods.py --pods 64 --steps 1500 > gpurun_out/r6_power_p64d.log 2>&1 & P=$!; sleep 9; sample 8; wait $P; tail -1 gpurun_out/r6_power_p64d.log | cut -c1-300 >> $OUT
echo "## 13B prefill 1024 tokens x 40 reps (k_gemm_b9)" >> $OUT
python tools/bench_prefill.py --shape 13B --n 1024 --reps 60 > gpurun_out/r6_power_p13.log 2>&1 & P=$!; sleep 12; sample 8; wait $P; tail -1 gpurun_out/r6_power_p13.log | cut -c1-300 >> $OUT
cat $OUT
