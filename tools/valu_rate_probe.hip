// tools/valu_rate_probe.hip — issue rate of the int8 -> bf16 conversion's instructions on gfx950: shader clocks per wave-instruction for
// independent streams of v_cvt_f32_i32 (SDWA byte select), v_cvt_f32_ubyteN, v_perm_b32, v_fma_f32, v_pk_fma_f32, with 1, 2 and 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ void k(unsigned long long* out, unsigned seed) {
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    float f0, f1, f2, f3, f4 = 1.f, f5 = 2.f, f6 = 3.f, f7 = 4.f;
    asm volatile("v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
        if (MODE == 0) { REP8(asm volatile("v_cvt_f32_i32_sdwa %0, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0\n v_cvt_f32_i32_sdwa %1, sext(%4) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1\n"
                                           "v_cvt_f32_i32_sdwa %2, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2\n v_cvt_f32_i32_sdwa %3, sext(%5) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3"
                                           : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(a0), "v"(a1));) }
        if (MODE == 1) { REP8(asm volatile("v_cvt_f32_ubyte0 %0, %4\n v_cvt_f32_ubyte1 %1, %4\n v_cvt_f32_ubyte2 %2, %5\n v_cvt_f32_ubyte3 %3, %5" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(a0), "v"(a1));) }
        if (MODE == 2) { REP8(asm volatile("v_perm_b32 %0, %4, %5, %6\n v_perm_b32 %1, %5, %4, %6\n v_perm_b32 %2, %4, %6, %7\n v_perm_b32 %3, %6, %5, %7" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));) }
        if (MODE == 3) { REP8(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(f4), "v"(f5));) }
        if (MODE == 4) { REP8(asm volatile("v_pk_fma_f32 %0, %2, %3, %0\n v_pk_fma_f32 %1, %2, %3, %1\n v_pk_fma_f32 %0, %3, %2, %0\n v_pk_fma_f32 %1, %3, %2, %1" : "+v"(*(double*)&f0), "+v"(*(double*)&f2) : "v"(*(double*)&f4), "v"(*(double*)&f6));) }
        if (MODE == 5) { REP8(asm volatile("v_cvt_f32_i32 %0, %4\n v_cvt_f32_i32 %1, %5\n v_cvt_f32_i32 %2, %4\n v_cvt_f32_i32 %3, %5" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(a0), "v"(a1));) }
        if (MODE == 6) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %4, %5\n v_cvt_pk_bf16_f32 %1, %5, %4\n v_cvt_pk_bf16_f32 %2, %4, %4\n v_cvt_pk_bf16_f32 %3, %5, %5" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(f4), "v"(f5));) }
        if (MODE == 7) { REP8(asm volatile("v_and_b32 %0, %4, %5\n v_xor_b32 %1, %5, %4\n v_lshrrev_b32 %2, 8, %4\n v_bfe_i32 %3, %5, 8, 8" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(a0), "v"(a1));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (f0 + f1 + f2 + f3 == 12345.f) out[1000] = 1;
}
template <int MODE> static int run(const char* name, unsigned long long* d) {
    printf("%-34s", name);
    for (int th = 256; th <= 1024; th += 256) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(th), 0, 0, d, 1u); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(th), 0, 0, d, 1u);
        CK(hipDeviceSynchronize());
        unsigned long long h[16]; CK(hipMemcpy(h, d + 16 * 100, sizeof h, hipMemcpyDeviceToHost));
        printf("  %d waves/SIMD: %.2f clk/instr/wave (%.2f per SIMD-instr)", th / 256, (double)h[0] / (64.0 * 32), (double)h[0] / (64.0 * 32) / (th / 256));
    }
    printf("\n"); return 0;
}
int main() {
    unsigned long long* d; CK(hipMalloc(&d, 8 * 4096 * 2));
    run<0>("v_cvt_f32_i32_sdwa (byte, sext)", d); run<5>("v_cvt_f32_i32", d); run<1>("v_cvt_f32_ubyteN", d); run<2>("v_perm_b32", d); run<3>("v_fma_f32", d); run<4>("v_pk_fma_f32", d);
    run<6>("v_cvt_pk_bf16_f32", d); run<7>("v_and/xor/lshr/bfe", d);
    return 0;
}
