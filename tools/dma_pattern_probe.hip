// tools/dma_pattern_probe.hip — what does the SHAPE of an LDS-DMA weight stream cost?  Round 6: k_stream_dma reads a 16-row tile in K-chunks of 64 / 128
// floats (one instruction = 4 rows x 256 B / 2 rows x 512 B) and reaches 5.3-5.4 TB/s on w1|w3 of 7B whatever the ring depth, the decode GEMV
// (one instruction = 1 KB of ONE row) 6.5.  Loader waves only, no matrix work, no activations:
//   smooth<KC, Q>: workgroup = 4 loader waves; a wave walks its share of (chunk, tile, row piece) instructions and keeps Q of them in flight
//                  (s_waitcnt vmcnt(Q - 1) in front of every issue, ring of Q KB of LDS per wave): no barrier at all
//   image<KC, NIMG, MAXT>: k_stream_dma's loader as it is: images of MAXT tiles x KC floats, wait for the oldest, s_barrier, request the next
// usage: dma_pattern_probe M K [reps = 20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000); }
template <int N>
__device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14)); }

template <int KC, int Q, int AUX>
__global__ __launch_bounds__(256) void k_smooth(const float* w, uint32_t M, uint32_t K, uint32_t* sink) {
    constexpr int NI = KC / 16;                       // 1 KB instructions per (chunk, tile)
    constexpr int NJ = NI / 4;                        // ... per wave
    static_assert(NJ >= 1, "chunk");
    constexpr int GR = KC >= 256 ? 64 : KC / 4;       // granules of one row inside an instruction
    constexpr int RPI = 64 / GR;                      // rows per instruction
    constexpr int SEG = KC >= 256 ? KC / 256 : 1;     // instructions per row
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t T = M >> 4, t0 = (uint32_t)(((uint64_t)blockIdx.x * T) / gridDim.x), t1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * T) / gridDim.x);
    uint32_t voff[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const uint32_t i = (uint32_t)j * 4 + wave;    // instruction of the unit
        const uint32_t row = KC >= 256 ? i / SEG : i * RPI + lane / GR, seg = KC >= 256 ? i % SEG : 0;
        const uint32_t gd = (uint32_t)lane % GR, gs = gd ^ (row & 15u & (GR - 1));
        voff[j] = (row * K + seg * 256u + gs * 4u) * 4u;
    }
    uint32_t n = 0;
    const uint32_t nch = K / KC;
    for (uint32_t c = 0; c < nch; ++c)
        for (uint32_t t = t0; t < t1; ++t) {
            const __amdgpu_buffer_rsrc_t rs = rsrc(w + (size_t)t * 16 * K);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                wait_vm<Q - 1>();
                __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(smem + ((size_t)wave * Q + n % Q) * 1024);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (int)voff[j], (int)(c * KC * 4u), 0, AUX);
                ++n;
            }
        }
    wait_vm<0>();
    if (sink == (uint32_t*)1) sink[threadIdx.x] = *(uint32_t*)(smem + threadIdx.x * 4);
}

template <int KC, int NIMG, int MAXT, int AUX>
__global__ __launch_bounds__(256) void k_image(const float* w, uint32_t M, uint32_t K, uint32_t* sink) {
    constexpr int GR = KC / 4, RPI = 64 / GR, ROWS = MAXT * 16, NIW = ROWS / RPI / 4;
    static_assert(KC == 64 || KC == 128, "chunk");
    constexpr int WAITN = NIW * (NIMG - 2) < 64 ? NIW * (NIMG - 2) : 63;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t T = M >> 4, t0 = (uint32_t)(((uint64_t)blockIdx.x * T) / gridDim.x), t1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * T) / gridDim.x), nt = t1 - t0;
    uint32_t voff[NIW];
    const float* base[NIW];
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const uint32_t q = (uint32_t)j * 4 + wave, rr = q * RPI + lane / GR, gd = lane % GR, gs = gd ^ (rr & 15u);
        uint32_t ti = (q * RPI) >> 4;
        ti = ti < nt ? ti : nt - 1;
        base[j] = w + (size_t)(t0 + ti) * 16 * K;
        voff[j] = ((rr & 15u) * K + gs * 4u) * 4u;
    }
    const uint32_t nch = K / KC;
    auto issue = [&](uint32_t ch) {
        const uint32_t cc = ch < nch ? ch : nch - 1;
        char* im = smem + (size_t)(ch % NIMG) * ROWS * KC * 4;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const __amdgpu_buffer_rsrc_t rs = rsrc(base[j]);
            __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(im + (size_t)(j * 4 + wave) * 1024);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (int)voff[j], (int)(cc * KC * 4u), 0, AUX);
        }
    };
#pragma unroll
    for (int c = 0; c < NIMG - 1; ++c) issue((uint32_t)c);
    for (uint32_t ch = 0; ch < nch; ++ch) {
        wait_vm<WAITN>();
        __builtin_amdgcn_s_barrier();
        issue(ch + NIMG - 1);
    }
    wait_vm<0>();
    if (sink == (uint32_t*)1) sink[threadIdx.x] = *(uint32_t*)(smem + threadIdx.x * 4);
}

int main(int argc, char** argv) {
    const uint32_t M = argc > 1 ? atoi(argv[1]) : 22016, K = argc > 2 ? atoi(argv[2]) : 4096;
    const int reps = argc > 3 ? atoi(argv[3]) : 20;
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); const int nCU = pr.multiProcessorCount;
    const size_t bytes = (size_t)M * K * 4;
    const int ncopy = (int)((1500ull << 20) / bytes) + 1;   // rotate over > 1.5 GB: nothing comes out of the Infinity Cache
    float* w; uint32_t* sink; CK(hipMalloc(&w, bytes * ncopy + (1 << 20))); CK(hipMemset(w, 0, bytes * ncopy)); CK(hipMalloc(&sink, 4096));
    printf("[M K = %u %u: %.1f MB, %d copies in rotation, %d CUs]\n", M, K, bytes / 1e6, ncopy, nCU);
    auto run = [&](auto kern, const char* name, int mult, size_t lds) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nCU * mult), dim3(256), lds, 0, (const float*)(w + (size_t)(i % ncopy) * M * K), M, K, sink);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(nCU * mult), dim3(256), lds, 0, (const float*)(w + (size_t)((i + 3) % ncopy) * M * K), M, K, sink);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-66s %7.2f us  %7.1f GB/s\n", name, ms * 1e3 / reps, bytes * (double)reps / ms / 1e6);
    };
    const uint32_t tiles_per_wg = ((M >> 4) + nCU - 1) / nCU;
#define SM(KC, Q, MULT) run(k_smooth<KC, Q, 2>, "smooth KC=" #KC " (" #Q " KB in flight per wave), " #MULT " wg/CU", MULT, (size_t)4 * Q * 1024)
    SM(64, 8, 1); SM(64, 16, 1); SM(64, 32, 1);
    SM(128, 8, 1); SM(128, 16, 1); SM(128, 32, 1);
    SM(256, 4, 1); SM(256, 8, 1); SM(256, 12, 1); SM(256, 16, 1); SM(256, 24, 1); SM(256, 32, 1);
    SM(512, 8, 1); SM(512, 16, 1); SM(512, 32, 1);
    SM(256, 8, 2); SM(256, 16, 2); SM(128, 16, 2);
    run(k_smooth<256, 16, 0>, "smooth KC=256 (16 KB per wave), temporal loads", 1, (size_t)4 * 16 * 1024);
#define IM(KC, NIMG, MAXT) if (tiles_per_wg <= MAXT) run(k_image<KC, NIMG, MAXT, 2>, "image KC=" #KC " x " #NIMG " images of " #MAXT " tiles (k_stream_dma's loader)", 1, (size_t)NIMG * MAXT * 16 * KC * 4)
    IM(128, 2, 6); IM(128, 3, 6); IM(64, 4, 6); IM(64, 4, 3); IM(128, 2, 3); IM(128, 3, 3); IM(64, 4, 1); IM(64, 5, 1); IM(128, 4, 1);
    printf("done\n");
    return 0;
}
