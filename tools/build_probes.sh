#!/bin/bash
# Builds the standalone checker of the stream GEMM and its timing-only traffic probes (see tools/stream_mm_check.hip) for gfx950.
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -o tools/stream_mm_check tools/stream_mm_check.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -o tools/q8b_probe tools/q8b_probe.hip || exit 1   # k_stream_q8b: checked runs + timeline (Q8B_TRACE) + ablation builds (-DQ8B_ABL=bits)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/valu_rate_probe tools/valu_rate_probe.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/valu_mfma_probe tools/valu_mfma_probe.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Wno-unused-result -o tools/mfma_clock_probe tools/mfma_clock_probe.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -Wno-unused-result -Illama.go_amd/csrc -Iinclude -DGEMM_CLOCK -o tools/gemm_probe_clock tools/gemm_probe.hip || exit 1
for p in pk_fma_probe sync_latency_probe; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-value -Wno-unused-result -o tools/$p tools/$p.hip || exit 1; done
ls -la tools/stream_mm_check*
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -Illama.go_amd/csrc -Iinclude -DB9_TRACE -o tools/gemm_b9_probe tools/gemm_b9_probe.hip || exit 1   # k_gemm_b9: checked runs, timing against k_gemm_glds, shader clock
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-result -Illama.go_amd/csrc -Iinclude -o tools/gemm_q8b3_probe tools/gemm_q8b3_probe.hip || exit 1   # k_gemm_q8b3: checked runs + timing against k_gemm_q8
