// tools/stream_probe.hip — does the ROW INTERLEAVE of a weight stream matter to HBM?  Not product code.
// A workgroup (4 waves) owns a contiguous block of rows of a [M][K] fp32 matrix, like the product kernels.  Each wave-instruction is a
// 16-byte-per-lane load; variant R = rows covered by one instruction:
//   R = 16: lane = (row l % 16, k-group l / 16)  -> 16 rows x 64 contiguous bytes   (A operand of v_mfma_f32_16x16x4_f32, k_skinny round 2)
//   R = 4 : lane = (row l % 4,  k-group l / 4)   ->  4 rows x 256 contiguous bytes  (A operand of v_mfma_f32_4x4x1_16B_f32)
//   R = 1 : lane = k-group l                     ->  1 row  x 1024 contiguous bytes (the decode GEMV)
// A wave walks the whole K of its rows (tile after tile), DEPTH loads in flight, and adds everything up (one VALU add per float4).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, int DEPTH>
__global__ __launch_bounds__(256) void k_probe(const float* __restrict__ w, float* __restrict__ out, uint32_t M, uint32_t K) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t nwg = gridDim.x;
    const uint32_t r0 = (uint32_t)(((uint64_t)blockIdx.x * M) / nwg), r1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * M) / nwg);
    constexpr int KG = 64 / R;                      // lanes per row in one instruction
    const uint32_t lr = lane % R, lk = lane / R;
    const uint32_t step = KG * 4;                   // floats of a row covered per instruction
    const uint32_t ntiles = (r1 - r0) / R, nsteps = K / step;
    f4 acc = {0, 0, 0, 0};
    // wave's tiles: wave, wave + 4, ...; items = (tile, step)
    const uint32_t mytiles = ntiles > (uint32_t)wave ? (ntiles - wave + 3) / 4 : 0;
    const uint32_t total = mytiles * nsteps;
    f4 ring[DEPTH];
    auto ptr = [&](uint32_t i) -> const f4* {
        const uint32_t t = i / nsteps, s = i - t * nsteps;
        const uint32_t row = r0 + (wave + 4 * t) * R + lr;
        return (const f4*)(w + (size_t)row * K + (size_t)s * step + lk * 4);
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d] = __builtin_nontemporal_load(ptr((uint32_t)d < total ? d : 0));
    for (uint32_t i0 = 0; i0 < total; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const f4 v = ring[d];
            acc += v;
            const uint32_t nx = i0 + d + DEPTH;
            ring[d] = __builtin_nontemporal_load(ptr(nx < total ? nx : 0));
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[tid] = acc.x;
}

static float* pool; static const size_t POOL = (size_t)6 << 30;
template <typename KT> static void run(const char* label, KT kern, uint32_t M, uint32_t K, int nCU, hipStream_t st, hipEvent_t e0, hipEvent_t e1, float* out) {
    const size_t bytes = (size_t)M * K * 4, nmat = POOL / bytes; const int iters = (int)(2e9 / bytes) + 8;
    auto launch = [&](int i) { hipLaunchKernelGGL(kern, dim3(nCU), dim3(256), 0, st, (const float*)(pool + (size_t)(i % nmat) * (bytes / 4)), out, M, K); };
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); const double us = ms * 1e3 / iters;
    printf("  %-46s %8.2f us  %7.1f GB/s\n", label, us, bytes / us / 1e3); CK(hipGetLastError());
}
int main() {
    CK(hipSetDevice(0)); hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); const int nCU = p.multiProcessorCount;
    CK(hipMalloc(&pool, POOL)); CK(hipMemset(pool, 0, POOL));
    float* out; CK(hipMalloc(&out, 4096));
    hipStream_t st; hipEvent_t e0, e1; CK(hipStreamCreate(&st)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct { const char* n; uint32_t M, K; } shapes[] = {{"w1w3 22016x4096", 22016, 4096}, {"wo 4096x4096", 4096, 4096}, {"w2 4096x11008", 4096, 11008}};
    for (auto& sh : shapes) {
        printf("[%s]\n", sh.n);
        run("16 rows x 64 B per instruction, 16 in flight", k_probe<16, 16>, sh.M, sh.K, nCU, st, e0, e1, out);
        run("16 rows x 64 B per instruction, 48 in flight", k_probe<16, 48>, sh.M, sh.K, nCU, st, e0, e1, out);
        run(" 4 rows x 256 B per instruction, 16 in flight", k_probe<4, 16>, sh.M, sh.K, nCU, st, e0, e1, out);
        run(" 4 rows x 256 B per instruction, 32 in flight", k_probe<4, 32>, sh.M, sh.K, nCU, st, e0, e1, out);
        run(" 1 row x 1024 B per instruction,  8 in flight", k_probe<1, 8>, sh.M, sh.K, nCU, st, e0, e1, out);
        run(" 1 row x 1024 B per instruction, 16 in flight", k_probe<1, 16>, sh.M, sh.K, nCU, st, e0, e1, out);
    }
    printf("done\n");
    return 0;
}
