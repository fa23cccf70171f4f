#!/bin/bash
# round 6: fp32 prompts of 129..256 tokens - two passes of the stream kernels per matrix (default up to 192) against the tile GEMM (LLAMAHIP_TWO_PASS_MAX=128), same box
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
OUT=gpurun_out/r6_two_pass.txt
: > $OUT
NS=128,129,136,144,160,176,192,200,208,224,240,256
echo "## tile GEMM (LLAMAHIP_TWO_PASS_MAX=128)" >> $OUT
LLAMAHIP_TWO_PASS_MAX=128 timeout 300 python tools/bench_ttft.py --ns $NS 2>&1 | tail -1 >> $OUT
echo "## two passes up to 256" >> $OUT
LLAMAHIP_TWO_PASS_MAX=256 timeout 300 python tools/bench_ttft.py --ns $NS 2>&1 | tail -1 >> $OUT
echo "## tile GEMM again" >> $OUT
LLAMAHIP_TWO_PASS_MAX=128 timeout 300 python tools/bench_ttft.py --ns $NS 2>&1 | tail -1 >> $OUT
echo "## default" >> $OUT
timeout 300 python tools/bench_ttft.py --ns $NS 2>&1 | tail -1 >> $OUT
cat $OUT
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -x -q -k "prefill_mfma_path or short_prompts_match or chunked_prefill or embeddings_of or reproducible" 2>&1 | tail -4
