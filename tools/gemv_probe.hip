// tools/gemv_probe.hip — design probe, not product code.
// Measures candidate fp32 GEMV layouts (y[M] = W[M,K] . x[K]) for the decode hot path on gfx950
// against a pure streaming-read ceiling, cycling through enough distinct matrices to defeat the
// 256 MiB Infinity Cache.  Build: hipcc --offload-arch=gfx950 -O3 -o gemv_probe gemv_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float float4_t __attribute__((ext_vector_type(4)));

template <bool NT> __device__ __forceinline__ float4_t ld4(const float4_t* p) {
    if (NT) return __builtin_nontemporal_load(p);
    return *p;
}

__device__ __forceinline__ float dpp_row_sum(float v) {
    // butterfly inside each row of 16 lanes, every lane ends with the row sum
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); // quad_perm 1,0,3,2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); // quad_perm 2,3,0,1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_row_sum(v);
    int iv = __builtin_bit_cast(int, v);
    float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (a + b) + (c + d);
}

// ---------------------------------------------------------------- V0: streaming read ceiling
template <bool NT, int U>
__global__ __launch_bounds__(256) void k_read(const float4_t* __restrict__ W, size_t n4, float* out) {
    size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float4_t acc = {0, 0, 0, 0};
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        float4_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld4<NT>(W + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n4; i += stride) acc += ld4<NT>(W + i);
    float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123.456f) out[0] = s;
}

// ---------------------------------------------------------------- V1: fat workgroup, K split across all threads
// 1024 threads = 16 waves, one workgroup per CU (dynamic LDS request keeps a second one out),
// thread t owns columns 4t..4t+3 (+4096j): x lives in KI float4 registers, rows streamed U at a time.
template <int KI, int U, bool NT, int TH = 1024>
__global__ __launch_bounds__(TH) void k_fat(const float4_t* __restrict__ W, const float4_t* __restrict__ x,
                                              float* __restrict__ y, int M, int K4) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* red = smem;  // [2][U][NW]
    constexpr int NW = TH / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwg = gridDim.x;
    const int r0 = (int)(((long long)blockIdx.x * M) / nwg), r1 = (int)(((long long)(blockIdx.x + 1) * M) / nwg);
    float4_t xr[KI];
    bool act[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        act[j] = tid + j * TH < K4;
        xr[j] = act[j] ? x[tid + j * TH] : float4_t{0, 0, 0, 0};
    }
    float4_t w[U][KI];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int j = 0; j < KI; ++j)
            w[u][j] = (r0 + u < r1 && act[j]) ? ld4<NT>(W + (size_t)(r0 + u) * K4 + tid + j * TH) : float4_t{0, 0, 0, 0};
    int buf = 0;
    for (int r = r0; r < r1; r += U) {
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float a = 0.f;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                float4_t c = w[u][j];
                a = fmaf(c.x, xr[j].x, a); a = fmaf(c.y, xr[j].y, a); a = fmaf(c.z, xr[j].z, a); a = fmaf(c.w, xr[j].w, a);
                int nr = r + U + u;
                if (nr < r1 && act[j]) w[u][j] = ld4<NT>(W + (size_t)nr * K4 + tid + j * TH);
            }
            acc[u] = a;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = wave_sum(acc[u]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) red[(buf * U + u) * NW + wave] = acc[u];
        }
        __syncthreads();
        if (tid < U && r + tid < r1) {
            const float* p = red + (buf * U + tid) * NW;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) s += p[k];
            y[r + tid] = s;
        }
        buf ^= 1;
    }
}

// ---------------------------------------------------------------- V2: wave per row, x staged in LDS
// 256 threads; each wave owns R consecutive rows at a time; UJ column steps in flight.
template <int R, int UJ, bool NT>
__global__ __launch_bounds__(256) void k_wave(const float4_t* __restrict__ W, const float4_t* __restrict__ x,
                                              float* __restrict__ y, int M, int K4) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float4_t* xs = (float4_t*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < K4; i += 256) xs[i] = x[i];
    __syncthreads();
    const int nwg = gridDim.x;
    // rows split evenly over workgroups, then R-row groups dealt round-robin to the 4 waves
    const int r0 = (int)(((long long)blockIdx.x * M) / nwg), r1 = (int)(((long long)(blockIdx.x + 1) * M) / nwg);
    const int nj = K4 / 64;  // K4 multiple of 64 assumed (4096 -> 16, 11008 -> 43)
    for (int row = r0 + wave * R; row < r1; row += 4 * R) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        int j = 0;
        for (; j + UJ <= nj; j += UJ) {
            float4_t wv[R][UJ];
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int u = 0; u < UJ; ++u)
                    wv[r][u] = (row + r < r1) ? ld4<NT>(W + (size_t)(row + r) * K4 + (j + u) * 64 + lane) : float4_t{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < UJ; ++u) {
                float4_t xv = xs[(j + u) * 64 + lane];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[r] = fmaf(wv[r][u].x, xv.x, acc[r]); acc[r] = fmaf(wv[r][u].y, xv.y, acc[r]);
                    acc[r] = fmaf(wv[r][u].z, xv.z, acc[r]); acc[r] = fmaf(wv[r][u].w, xv.w, acc[r]);
                }
            }
        }
        for (; j < nj; ++j) {
            float4_t xv = xs[j * 64 + lane];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (row + r < r1) {
                    float4_t c = ld4<NT>(W + (size_t)(row + r) * K4 + j * 64 + lane);
                    acc[r] = fmaf(c.x, xv.x, acc[r]); acc[r] = fmaf(c.y, xv.y, acc[r]);
                    acc[r] = fmaf(c.z, xv.z, acc[r]); acc[r] = fmaf(c.w, xv.w, acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = wave_sum(acc[r]);
            if (lane == 0 && row + r < r1) y[row + r] = s;
        }
    }
}

struct Shape { const char* name; int M, K; };

static double ref_row(const std::vector<float>& w, const std::vector<float>& x, int K) {
    double s = 0; for (int k = 0; k < K; ++k) s += (double)w[k] * x[k]; return s;
}

int main(int argc, char** argv) {
    int dev = 0; CK(hipSetDevice(dev));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, dev));
    int nCU = p.multiProcessorCount;
    printf("device %s arch %s CUs %d clock %d kHz memclk %d kHz bus %d bits  L2 %d  totalGlobalMem %.1f GB maxSmemPerBlock %zu\n",
           p.name, p.gcnArchName, nCU, p.clockRate, p.memoryClockRate, p.memoryBusWidth, p.l2CacheSize,
           p.totalGlobalMem / 1e9, p.sharedMemPerBlock);
    const size_t POOL = (size_t)6 << 30;  // 6 GiB pool of weights to cycle through
    float* pool; CK(hipMalloc(&pool, POOL));
    {   // fill pool with small pseudo-random values (host LCG, 64 MiB pattern replicated)
        size_t pat = (size_t)16 << 20; std::vector<float> h(pat);
        unsigned s = 12345; for (size_t i = 0; i < pat; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
        for (size_t off = 0; off < POOL; off += pat * 4) CK(hipMemcpy((char*)pool + off, h.data(), pat * 4, hipMemcpyHostToDevice));
    }
    float *x, *y; CK(hipMalloc(&x, 65536 * 4)); CK(hipMalloc(&y, 65536 * 4));
    std::vector<float> hx(65536); { unsigned s = 777; for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); } }
    CK(hipMemcpy(x, hx.data(), 65536 * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    Shape shapes[] = {{"w1w3 22016x4096", 22016, 4096}, {"qkv 12288x4096", 12288, 4096}, {"wo 4096x4096", 4096, 4096},
                      {"w2 4096x11008", 4096, 11008}, {"lmhead 32000x4096", 32000, 4096}};

    auto timeit = [&](const char* label, size_t bytes_per_launch, int iters, auto launch) {
        size_t nmat = POOL / bytes_per_launch; if (nmat < 1) nmat = 1;
        for (int i = 0; i < 3; ++i) launch(pool + (i % nmat) * (bytes_per_launch / 4));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) launch(pool + ((size_t)(i % nmat)) * (bytes_per_launch / 4));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double us = ms * 1e3 / iters;
        printf("  %-44s %8.2f us  %7.1f GB/s\n", label, us, bytes_per_launch / us / 1e3);
        hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("   !! %s\n", hipGetErrorString(e));
    };

    // ---- V0 ceiling on a 344 MiB slab
    {
        size_t bytes = (size_t)22016 * 4096 * 4; size_t n4 = bytes / 16;
        printf("[V0] streaming read, %zu MiB per launch\n", bytes >> 20);
        for (int g : {nCU * 4, nCU * 8, nCU * 16, nCU * 32}) {
            char l[96];
            snprintf(l, 96, "read nt=0 U=8 grid=%d", g); timeit(l, bytes, 40, [&](float* w) { k_read<false, 8><<<g, 256, 0, st>>>((const float4_t*)w, n4, y); });
            snprintf(l, 96, "read nt=1 U=8 grid=%d", g); timeit(l, bytes, 40, [&](float* w) { k_read<true, 8><<<g, 256, 0, st>>>((const float4_t*)w, n4, y); });
            snprintf(l, 96, "read nt=1 U=4 grid=%d", g); timeit(l, bytes, 40, [&](float* w) { k_read<true, 4><<<g, 256, 0, st>>>((const float4_t*)w, n4, y); });
        }
    }
    // ---- correctness helper
    auto check = [&](const char* label, int M, int K, float* w) {
        std::vector<float> hy(M); CK(hipMemcpy(hy.data(), y, M * 4, hipMemcpyDeviceToHost));
        std::vector<float> hw((size_t)K);
        double maxrel = 0;
        for (int r : {0, 1, M / 2 + 1, M - 2, M - 1}) {
            CK(hipMemcpy(hw.data(), w + (size_t)r * K, K * 4, hipMemcpyDeviceToHost));
            double ref = ref_row(hw, hx, K);
            double rel = fabs(hy[r] - ref) / (fabs(ref) + 1e-3);
            if (rel > maxrel) maxrel = rel;
        }
        printf("  check %-38s max rel err %.2e %s\n", label, maxrel, maxrel < 1e-3 ? "ok" : "FAIL");
    };

    for (auto& s : shapes) {
        size_t bytes = (size_t)s.M * s.K * 4; int K4 = s.K / 4;
        int iters = (int)(4e9 / bytes) + 8;
        printf("[%s] %zu MiB per launch, iters %d\n", s.name, bytes >> 20, iters);
        // V1 fat
        auto fat = [&](auto kern, const char* label, int U, int TH = 1024, int perCU = 1) {
            size_t lds = (perCU == 1 ? 96 : perCU == 2 ? 64 : 36) * 1024;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            CK(hipMemset(y, 0, s.M * 4));
            kern<<<nCU * perCU, TH, lds, st>>>((const float4_t*)pool, (const float4_t*)x, y, s.M, K4);
            CK(hipStreamSynchronize(st)); check(label, s.M, s.K, pool);
            timeit(label, bytes, iters, [&](float* w) { kern<<<nCU * perCU, TH, lds, st>>>((const float4_t*)w, (const float4_t*)x, y, s.M, K4); });
        };
        if (s.K == 4096) {
            fat(k_fat<1, 2, true>, "fat KI=1 U=2 nt", 2);
            fat(k_fat<1, 3, true>, "fat KI=1 U=3 nt", 3);
            fat(k_fat<1, 4, true>, "fat KI=1 U=4 nt", 4);
            fat(k_fat<1, 6, true>, "fat KI=1 U=6 nt", 6);
            fat(k_fat<2, 2, true, 512>, "fat512x2 KI=2 U=2 nt", 2, 512, 2);
            fat(k_fat<2, 4, true, 512>, "fat512x2 KI=2 U=4 nt", 4, 512, 2);
            fat(k_fat<4, 1, true, 256>, "fat256x4 KI=4 U=1 nt", 1, 256, 4);
            fat(k_fat<4, 2, true, 256>, "fat256x4 KI=4 U=2 nt", 2, 256, 4);
        } else {
            fat(k_fat<3, 1, true>, "fat KI=3 U=1 nt", 1);
            fat(k_fat<3, 2, true>, "fat KI=3 U=2 nt", 2);
            fat(k_fat<3, 3, true>, "fat KI=3 U=3 nt", 3);
            fat(k_fat<6, 1, true, 512>, "fat512x2 KI=6 U=1 nt", 1, 512, 2);
            fat(k_fat<6, 2, true, 512>, "fat512x2 KI=6 U=2 nt", 2, 512, 2);
        }
        // V2 wave-per-row
        auto wav = [&](auto kern, const char* label, int wgPerCU) {
            size_t lds = (size_t)s.K * 4;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int g = nCU * wgPerCU;
            CK(hipMemset(y, 0, s.M * 4));
            kern<<<g, 256, lds, st>>>((const float4_t*)pool, (const float4_t*)x, y, s.M, K4);
            CK(hipStreamSynchronize(st)); check(label, s.M, s.K, pool);
            char l[96]; snprintf(l, 96, "%s wg/CU=%d", label, wgPerCU);
            timeit(l, bytes, iters, [&](float* w) { kern<<<g, 256, lds, st>>>((const float4_t*)w, (const float4_t*)x, y, s.M, K4); });
        };
        int maxwg = s.K == 4096 ? 8 : 3;
        for (int wg : {2, 4, maxwg}) {
            if (wg > maxwg) continue;
            wav(k_wave<1, 8, true>, "wave R=1 UJ=8 nt", wg);
            wav(k_wave<1, 4, true>, "wave R=1 UJ=4 nt", wg);
        }

    }
    printf("done\n");
    return 0;
}
