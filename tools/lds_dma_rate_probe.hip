// tools/lds_dma_rate_probe.hip — how fast can a CU fill its LDS (or its registers) from L2-RESIDENT data?  Round 6: k_stream_b9's launches take
// (weight bytes + 256 CUs x plane bytes) / 6.5 TB/s whatever the arithmetic - is the global -> LDS path of a CU limited to ~12 B/clk even for L2 hits?
// Every workgroup reads the SAME `bytes` (L2 / MALL resident after the first touch) `iters` times:
//   mode 0: buffer_load_dwordx4 ... lds (LDS-DMA), mode 1: global_load_dwordx4 into registers (xor-reduced), mode 2: mode 1 + ds_write_b128
// usage: lds_dma_rate_probe [bytes per pass = 24576] [iters = 2000] [threads = 1024]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef int i4v __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(1024) void k_fill(const char* src, uint32_t bytes, uint32_t iters, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t tid = threadIdx.x, nth = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t per_pass = nth * 16;                 // bytes one instruction of every wave moves
    const uint32_t npieces = bytes / per_pass;          // instructions per wave and pass
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    u4 acc = {0, 0, 0, 0};
    for (uint32_t it = 0; it < iters; ++it) {
        for (uint32_t p = 0; p < npieces; ++p) {
            const uint32_t off = p * per_pass + wave * 1024;
            if (MODE == 0) {
                const uint64_t b = (uint64_t)src;
                const i4v rs = {(int)(uint32_t)b, (int)((uint32_t)(b >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + off), so = __builtin_amdgcn_readfirstlane(off);
                const uint32_t vo = lane * 16;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(vo), "s"(rs), "s"(so) : "memory", "m0");
            } else {
                const u4 v = *(const u4*)(src + off + lane * 16);
                if (MODE == 2) *(u4*)(smem + off + lane * 16) = v;
                else acc ^= v;
            }
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 2) __syncthreads();
    }
    if (MODE != 0 && acc.x == 0x12345678u) sink[tid] = acc.y ^ acc.z ^ acc.w;
    if (MODE != 1 && sink == (uint32_t*)1) sink[tid] = *(uint32_t*)(smem + tid * 4);
}
int main(int argc, char** argv) {
    const uint32_t bytes = argc > 1 ? atoi(argv[1]) : 24576, iters = argc > 2 ? atoi(argv[2]) : 2000, th = argc > 3 ? atoi(argv[3]) : 1024;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); const int nCU = pr.multiProcessorCount;
    char* src; uint32_t* sink; CK(hipMalloc(&src, bytes + 65536)); CK(hipMemset(src, 1, bytes + 65536)); CK(hipMalloc(&sink, 4096 * 4));
    const size_t lds = 96 * 1024;   // one workgroup per CU
    auto run = [&](auto kern, const char* name) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(nCU), dim3(th), lds, 0, (const char*)src, bytes, 10u, sink);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(nCU), dim3(th), lds, 0, (const char*)src, bytes, iters, sink);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double tot = (double)nCU * bytes * iters;
        printf("%-44s %u B x %u passes, %u threads: %.1f us, %.2f TB/s aggregate = %.1f GB/s per CU = %.1f B/clk/CU at 2.1 GHz\n", name, bytes, iters, th, ms * 1e3, tot / ms / 1e9, tot / ms / 1e6 / nCU,
               tot / ms / 1e6 / nCU / 2.1);
    };
    run(k_fill<0>, "LDS-DMA (buffer_load_dwordx4 ... lds)");
    run(k_fill<1>, "global_load_dwordx4 -> registers");
    run(k_fill<2>, "global_load_dwordx4 -> ds_write_b128");
    return 0;
}
