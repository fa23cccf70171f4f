// tools/mfma_peak.hip — what the fp32 MFMA pipe delivers on this chip with nothing else in the way: every wave runs a loop of
// independent v_mfma_f32_32x32x2_f32 on register operands.  Variants: waves per SIMD (1, 2, 4) and accumulators per wave (1..5).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak tools/mfma_peak.hip ; run: ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float a0, float b0) {
    f16v acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NACC; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    if (s == 123.456f) out[0] = s;
}

template <int NACC>
static void run(int wgs_per_cu, int iters) {
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * wgs_per_cu;
    k_mfma<NACC><<<grid, 256>>>(out, 10, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_mfma<NACC><<<grid, 256>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("acc/wave %d  waves/SIMD %d  %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", NACC, wgs_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<1>(w, 20000 / w);
        run<2>(w, 10000 / w);
        run<4>(w, 5000 / w);
        run<5>(w, 4000 / w);
    }
    // long run: clocks settle under sustained MFMA load
    run<4>(2, 100000);
    return 0;
}
