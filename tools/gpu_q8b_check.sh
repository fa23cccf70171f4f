#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/bench_pods.py --int8 --pods 1,5,8,16,32,48,64 --steps 32 > gpurun_out/q8b_pods.json 2> gpurun_out/q8b_pods.err; tail -3 gpurun_out/q8b_pods.err; cat gpurun_out/q8b_pods.json
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -x -q -k "int8 or q8" 2>&1 | tail -5 > gpurun_out/q8b_tests.txt
cat gpurun_out/q8b_tests.txt
