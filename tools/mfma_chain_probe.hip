// tools/mfma_chain_probe.hip — v_mfma_f32_16x16x32_bf16 from ONE wave per SIMD: shader clocks per MFMA with NA independent accumulators in rotation
// (how far apart must two MFMAs on the same accumulator stand?), bare and with F vector instructions (v_and / v_sub pairs) behind every MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4m __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int NA, int F>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, uint32_t iters) {
    f4m acc[NA];
    for (int i = 0; i < NA; ++i) acc[i] = f4m{0.f, 0.f, 0.f, 0.f};
    u4 a = {threadIdx.x, 1, 2, 3}, b = {4, 5, threadIdx.x, 7};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)threadIdx.x + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    const int q = (r * NA * F + i * F + f) & 7;
                    const uint32_t h = __builtin_bit_cast(uint32_t, v[q]) & 0xffff0000u;
                    v[q] = (f & 1) ? __fsub_rn(v[q], __builtin_bit_cast(float, h)) : __builtin_bit_cast(float, h | 0x3f800000u);
                    asm volatile("" : "+v"(v[q]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < NA; ++i) s += acc[i][0]; for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.f) sink[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
template <int NA, int F> static void run(int nCU) {
    unsigned long long* d; float* sink; CK(hipMalloc(&d, 8)); CK(hipMalloc(&sink, 4096));
    const uint32_t iters = 2000;
    hipLaunchKernelGGL((k<NA, F>), dim3(nCU), dim3(256), 0, 0, d, sink, iters); CK(hipDeviceSynchronize());
    unsigned long long t; CK(hipMemcpy(&t, d, 8, hipMemcpyDeviceToHost));
    printf("%d accumulators, %d vector instructions per MFMA: %.1f clocks per MFMA\n", NA, F, (double)t / (iters * 8.0 * NA));
}
int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); const int nCU = p.multiProcessorCount;
    run<1, 0>(nCU); run<2, 0>(nCU); run<3, 0>(nCU); run<4, 0>(nCU); run<6, 0>(nCU); run<8, 0>(nCU);
    run<4, 1>(nCU); run<4, 2>(nCU); run<4, 3>(nCU); run<4, 4>(nCU); run<2, 2>(nCU); run<8, 2>(nCU);
    return 0;
}
