#!/bin/bash
# Builds the standalone checker of the stream GEMM and its timing-only traffic probes (see tools/stream_mm_check.hip) for gfx950.
cd "$(dirname "$0")/.."
for p in "" 1 2 4 8 9 16; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 ${p:+-DSTREAM_PROBE=$p} -o tools/stream_mm_check${p:+_p$p} tools/stream_mm_check.hip || exit 1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/valu_mfma_probe tools/valu_mfma_probe.hip || exit 1
# the loader on buffer loads (STREAM_BUFFER_LOADS, computes correct results: run WITH the check)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSTREAM_BUFFER_LOADS=1 -o tools/stream_mm_check_buf tools/stream_mm_check.hip || exit 1
# ... and with the loaders sleeping 6 x 64 clocks behind every chunk barrier (STREAM_LOADER_SLEEP): LDS queue goes to the operand reads first
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSTREAM_BUFFER_LOADS=1 -DSTREAM_LOADER_SLEEP=6 -o tools/stream_mm_check_buf_sleep6 tools/stream_mm_check.hip || exit 1
# ... and with the MFMA waves paced (s_nop behind every MFMA, STREAM_MFMA_PACE): with and without the buffer loads
for n in 6 8 9 10 11; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSTREAM_MFMA_PACE=$n -o tools/stream_mm_check_pace$n tools/stream_mm_check.hip || exit 1
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSTREAM_BUFFER_LOADS=1 -DSTREAM_MFMA_PACE=9 -o tools/stream_mm_check_buf_pace9 tools/stream_mm_check.hip || exit 1
ls -la tools/stream_mm_check*
