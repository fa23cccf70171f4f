// tools/mfma_clock_probe.hip — what shader clock does the chip hold while every SIMD issues fp32 MFMAs back to back, and what fp32 matrix rate is that?
// One workgroup of 4 waves per CU (x `wgs_per_cu`), each wave loops over 8 independent v_mfma_f32_32x32x2_f32 accumulators (no operand traffic at all:
// this is the ceiling a GEMM's inner loop can approach).  A wave of the middle workgroup stamps s_memtime (shader clocks) and s_memrealtime (100 MHz)
// around its loop; the host times the launch with HIP events.  Prints GHz, TFLOP/s and the fraction of the 157.3 TFLOP/s a 2.4 GHz clock would give.
// usage: tools/mfma_clock_probe [iters=200000] [wgs_per_cu=1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v16f __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma_burn(unsigned iters, float* sink, unsigned long long* clk) {
    v16f acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    // operands: pseudo-random per lane (what the multipliers toggle moves the power, and with it the clock)
    unsigned h = (threadIdx.x + blockIdx.x * 256u) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float a = (float)((int)(h >> 8) - (1 << 23)) / (float)(1 << 23), b = (float)((int)((h * 3266489917u) >> 8) - (1 << 23)) / (float)(1 << 23);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[i][e];
    if (s == 12345.678f) sink[0] = s;   // keeps the loop alive
    if (blockIdx.x == gridDim.x / 2 && threadIdx.x == 64) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}

int main(int argc, char** argv) {
    const unsigned iters = argc > 1 ? (unsigned)atoi(argv[1]) : 200000u;
    const int per_cu = argc > 2 ? atoi(argv[2]) : 1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) != hipSuccess) { fprintf(stderr, "no device\n"); return 1; }
    const int cus = prop.multiProcessorCount;
    float* sink; unsigned long long* clk;
    hipMalloc(&sink, 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mfma_burn, dim3(cus * per_cu), dim3(256), 0, 0, iters, sink, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);              // shader clocks per ns (s_memrealtime ticks at 100 MHz)
        const double flop = 2.0 * 32 * 32 * 2 * 8.0 * iters * 4.0 * cus * per_cu;   // per MFMA 32x32x2 = 2048 MACs
        const double tf = flop / (ms * 1e-3) / 1e12;
        const double cyc_per_mfma = (double)h[0] / (8.0 * iters) / per_cu;
        printf("{\"rep\": %d, \"cus\": %d, \"wgs_per_cu\": %d, \"ms\": %.3f, \"shader_GHz\": %.4f, \"clocks_per_mfma_per_simd\": %.2f, \"TFLOPs\": %.2f, \"of_157.3\": %.4f, \"peak_at_this_clock_TFLOPs\": %.2f}\n",
               rep, cus, per_cu, ms, ghz, cyc_per_mfma, tf, tf / 157.3, 157.3 * ghz / 2.4);
    }
    return 0;
}
