// tools/pk_fma_probe.hip — issue rate of v_pk_fma_f32 against v_fma_f32 on gfx950: one wave per SIMD, 32 independent accumulator chains, shader clocks
// per instruction from s_memtime.  usage: tools/pk_fma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(unsigned iters, float* sink, unsigned long long* clk, float a0) {
    f2 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f2{(float)i, (float)threadIdx.x};
    f2 a = f2{a0, a0 * 0.5f}, b = f2{1.0f + a0, 1.0f - a0};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (unsigned it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (PK) acc[i] = __builtin_elementwise_fma(a, acc[i], b);
            else acc[i].x = fmaf(a.x, acc[i].x, b.x);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i].x + acc[i].y;
    if (s == 1234.5f) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = c1 - c0;
}
int main() {
    float* sink; unsigned long long* clk; hipMalloc(&sink, 4); hipMalloc(&clk, 8);
    const unsigned iters = 20000;
    for (int pk = 0; pk < 2; ++pk) {
        for (int rep = 0; rep < 2; ++rep) {
            if (pk) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, iters, sink, clk, 0.001f);
            else hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, iters, sink, clk, 0.001f);
            hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, clk, 8, hipMemcpyDeviceToHost);
        printf("%s: %.2f shader clocks per instruction and wave (one wave per SIMD, 32 independent chains)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", (double)h / (32.0 * iters));
    }
    return 0;
}
