// tools/persist_probe.hip — does a RESIDENT workgroup-per-CU kernel with grid barriers + LDS-parked next-phase rows beat one kernel per
// GEMV?  Chain of the four big 7B decode GEMVs per layer (12288x4096, 4096x4096, 22016x4096, 4096x11008; each output feeds the next,
// + residual), 32 layers = 25.9 GB of distinct weights, built from the product blocks in csrc/kernels_persist.h.  Not product code.
//   baseline A: the product kernels (k_gemv_sa, product launch shapes) replayed as one hipGraph
//   baseline B: k_gemv_sa with the persistent kernel's own shapes (512 threads) -> results must match the persistent kernel BIT FOR BIT
//   persistent: coherence mode (fences / uncached buffers / scoped accesses) x poll path (vector / scalar) x parked bytes
// Every spin is time-bounded: a variant that cannot synchronise reports "TIMEOUT", it cannot hang the GPU.
#include "kernels_persist.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
#include <functional>
#include <algorithm>
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct ProbePhase {
    const float* w; uint32_t M, K;
    const float* x; const float* resid; float* y;
};

__global__ void k_fill(float* p, size_t n, uint32_t seed, float scale) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull * (seed + 1));
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = ((float)(int)((z >> 40) & 0xFFFFFF) - 8388608.0f) * (1.0f / 8388608.0f) * scale;
    }
}

// Phases come in layers of four with a fixed shape sequence (K <= 2048*4 three times, then the long-K one), like the product's
// layer schedule: each phase is its own straight-line instantiation (a run-time choice between instantiations makes the compiler
// merge their tails, and its memory-counter bookkeeping then guards the next phase's first register writes with a full drain).
// Phases come in layers of four with a fixed shape sequence (K <= 2048*4 three times, then the long-K one), like the product's
// layer schedule: each phase is its own straight-line instantiation.  NP2 / NP6: rows parked across the barrier (K=4096 / K=11008 phases).
template <int XM, int POLL, int UD, int NP2, int E2, int NP6, int E6>
__global__ __launch_bounds__(PTH) void k_persist_probe(const ProbePhase* __restrict__ phases, uint32_t nphases, const PersistCtl c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t wg = blockIdx.x, nwg = gridDim.x;
    const unsigned long long base = persist_base(c, nwg);
    bool aborted = false;
    auto args = [&](uint32_t ph) {
        const ProbePhase d = phases[ph < nphases ? ph : nphases - 1];
        GemvArgs a = {};
        a.w[0] = d.w; a.M = d.M; a.K = d.K; a.x = d.x; a.resid = d.resid; a.y = d.y;
        return a;
    };
    f4 pk2[NP2 > 0 ? NP2 : 1][2], pk6[NP6 > 0 ? NP6 : 1][6];
    {
        GemvArgs a0 = args(0);
        persist_park<2, NP2, 0, NP2, MAP_SINGLE>(a0, c, wg, nwg, pk2);
    }
    for (uint32_t ph = 0; ph < nphases; ph += 4) {
        GemvArgs a0 = args(ph), a1 = args(ph + 1), a2 = args(ph + 2), a3 = args(ph + 3), an = args(ph + 4);
        persist_gemv<XM, 2, UD, NP2, PRO_PLAIN, EPI_RESID, MAP_SINGLE>(a0, c, wg, nwg, smem, pk2, [&] { persist_park<2, NP2, 0, E2, MAP_SINGLE>(a1, c, wg, nwg, pk2); });
        persist_boundary<XM, POLL, 2, NP2, E2, MAP_SINGLE>(c, a1, wg, nwg, base, ph, &aborted, pk2);
        persist_gemv<XM, 2, UD, NP2, PRO_PLAIN, EPI_RESID, MAP_SINGLE>(a1, c, wg, nwg, smem, pk2, [&] { persist_park<2, NP2, 0, E2, MAP_SINGLE>(a2, c, wg, nwg, pk2); });
        persist_boundary<XM, POLL, 2, NP2, E2, MAP_SINGLE>(c, a2, wg, nwg, base, ph + 1, &aborted, pk2);
        persist_gemv<XM, 2, UD, NP2, PRO_PLAIN, EPI_RESID, MAP_SINGLE>(a2, c, wg, nwg, smem, pk2, [&] { persist_park<6, NP6, 0, E6, MAP_SINGLE>(a3, c, wg, nwg, pk6); });
        persist_boundary<XM, POLL, 6, NP6, E6, MAP_SINGLE>(c, a3, wg, nwg, base, ph + 2, &aborted, pk6);
        if (ph + 4 < nphases) {
            persist_gemv<XM, 6, 1, NP6, PRO_PLAIN, EPI_RESID, MAP_SINGLE>(a3, c, wg, nwg, smem, pk6, [&] { persist_park<2, NP2, 0, E2, MAP_SINGLE>(an, c, wg, nwg, pk2); });
            persist_boundary<XM, POLL, 2, NP2, E2, MAP_SINGLE>(c, an, wg, nwg, base, ph + 3, &aborted, pk2);
        } else {
            persist_gemv<XM, 6, 1, NP6, PRO_PLAIN, EPI_RESID, MAP_SINGLE>(a3, c, wg, nwg, smem, pk6);
        }
    }
    persist_report(c, aborted);
}


// ---- attention || wo: ONE launch, heterogeneous blocks.  Blocks [0, H) stand in for the attention heads (latency-bound, ~5 us: here a timed
// spin + 128 scoped stores + arrive), blocks [H, H + #CU) are the wo GEMV: they request ALL their weight rows first (weights do not depend on
// the attention output), then wait for the H arrivals, fetch x and finish.  Lower block ids are dispatched first, so the heads can never
// be starved by waiting GEMV blocks.  Self-resetting counters: the last GEMV block to leave zeroes them for the next launch.
struct AwSync { unsigned int* cnt; unsigned int* done; };
template <int NP, int U>
__global__ __launch_bounds__(PTH) void k_aw_probe(GemvArgs a, PersistCtl c, AwSync sy, uint32_t H, uint32_t spin_ticks, float* attn_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x < H) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(2);
        if (threadIdx.x < 128) stx1<XM_SCOPED>(attn_out + blockIdx.x * 128 + threadIdx.x, 0.001f * (float)(threadIdx.x + blockIdx.x));
        wait_vmcnt<0>();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(sy.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const uint32_t wg = blockIdx.x - H, nwg = gridDim.x - H;
    f4 pk[NP > 0 ? NP : 1][2];
    persist_park<2, NP, 0, NP, MAP_SINGLE>(a, c, wg, nwg, pk);
    if (threadIdx.x < 64) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(sy.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < H) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > (uint64_t)c.timeout_ticks) { if (threadIdx.x == 0) __hip_atomic_store(c.err, 7u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    persist_gemv<XM_SCOPED, 2, U, NP, PRO_PLAIN, EPI_RESID, MAP_SINGLE>(a, c, wg, nwg, smem, pk);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int old = __hip_atomic_fetch_add(sy.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nwg - 1) {
            __hip_atomic_store(sy.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sy.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ __launch_bounds__(PTH) void k_spin_heads(uint32_t spin_ticks, float* attn_out) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x < 128) attn_out[blockIdx.x * 128 + threadIdx.x] = 0.001f * (float)(threadIdx.x + blockIdx.x);
}


// two launches that overlap through a forked capture: heads on stream 2, the waiting wo GEMV on stream 1 (gemv_only: every block is a GEMV block)
template <int NP, int U>
__global__ __launch_bounds__(PTH) void k_wo_wait(GemvArgs a, PersistCtl c, AwSync sy, uint32_t H) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t wg = blockIdx.x, nwg = gridDim.x;
    f4 pk[NP > 0 ? NP : 1][2];
    persist_park<2, NP, 0, NP, MAP_SINGLE>(a, c, wg, nwg, pk);
    if (threadIdx.x < 64) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__hip_atomic_load(sy.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < H) {
            if (__builtin_amdgcn_s_memrealtime() - t0 > (uint64_t)c.timeout_ticks) { if (threadIdx.x == 0) __hip_atomic_store(c.err, 9u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    persist_gemv<XM_SCOPED, 2, U, NP, PRO_PLAIN, EPI_RESID, MAP_SINGLE>(a, c, wg, nwg, smem, pk);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int old = __hip_atomic_fetch_add(sy.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == nwg - 1) {
            __hip_atomic_store(sy.cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sy.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
__global__ __launch_bounds__(PTH) void k_heads_arrive(uint32_t spin_ticks, float* attn_out, AwSync sy) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(2);
    if (threadIdx.x < 128) stx1<XM_SCOPED>(attn_out + blockIdx.x * 128 + threadIdx.x, 0.001f * (float)(threadIdx.x + blockIdx.x));
    wait_vmcnt<0>();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(sy.cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static int nCU;
static hipStream_t st;
static hipEvent_t e0, e1;

static float time_ms(const std::function<void()>& f, int reps) {
    f();
    CK(hipStreamSynchronize(st));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, st));
        f();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return best;
}

int main(int argc, char** argv) {
    const int L = argc > 1 ? atoi(argv[1]) : 32;
    CK(hipSetDevice(0)); hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); nCU = p.multiProcessorCount;
    printf("device %s CUs %d, layers %d\n", p.gcnArchName, nCU, L);
    CK(hipStreamCreate(&st)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t d = 4096, F = 11008;
    const uint32_t Ms[4] = {3 * d, d, 2 * F, d}, Ks[4] = {d, d, d, F};
    size_t per_layer = 0; for (int i = 0; i < 4; ++i) per_layer += (size_t)Ms[i] * Ks[i];
    float* W; CK(hipMalloc(&W, per_layer * L * 4));
    {   // one fill per matrix so the scale follows its fan-in (outputs stay O(1) through the chain)
        size_t off = 0;
        for (int l = 0; l < L; ++l) for (int i = 0; i < 4; ++i) {
            const size_t n = (size_t)Ms[i] * Ks[i];
            hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, W + off, n, (uint32_t)(l * 4 + i), sqrtf(3.0f / Ks[i]) * (i == 3 ? 0.5f : 1.0f));
            off += n;
        }
        CK(hipStreamSynchronize(st));
    }
    // exchange buffers in three flavours: normal, uncached, fine-grained
    struct Ex { float *v, *ya, *yb, *yc; unsigned long long* count; uint32_t* err; const char* name; bool ok; };
    auto alloc_ex = [&](unsigned flags, const char* name) {
        Ex e = {}; e.name = name; e.ok = true;
        auto al = [&](void** ptr, size_t bytes) {
            hipError_t r = flags == 0xFFFF ? hipMalloc(ptr, bytes) : hipExtMallocWithFlags(ptr, bytes, flags);
            if (r != hipSuccess) { e.ok = false; *ptr = nullptr; (void)hipGetLastError(); }
        };
        al((void**)&e.v, d * 4); al((void**)&e.ya, 3 * d * 4); al((void**)&e.yb, d * 4); al((void**)&e.yc, 2 * F * 4); al((void**)&e.count, 64); al((void**)&e.err, 64);
        if (e.ok) { CK(hipMemset(e.count, 0, 64)); CK(hipMemset(e.err, 0, 64)); }
        printf("exchange buffers %-12s %s\n", name, e.ok ? "allocated" : "NOT AVAILABLE");
        return e;
    };
    Ex exN = alloc_ex(0xFFFF, "normal"), exU = alloc_ex(hipDeviceMallocUncached, "uncached"), exF = alloc_ex(hipDeviceMallocFinegrained, "finegrained");
    float *zeros, *v0, *dummy;
    CK(hipMalloc(&zeros, 2 * F * 4)); CK(hipMemset(zeros, 0, 2 * F * 4));
    CK(hipMalloc(&dummy, 65536)); CK(hipMemset(dummy, 0, 65536));
    CK(hipMalloc(&v0, d * 4)); hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, st, v0, (size_t)d, 777u, 1.0f); CK(hipStreamSynchronize(st));

    auto make_phases = [&](const Ex& e, std::vector<ProbePhase>& ph) {
        ph.clear(); size_t off = 0;
        for (int l = 0; l < L; ++l) {
            const float* w[4]; for (int i = 0; i < 4; ++i) { w[i] = W + off; off += (size_t)Ms[i] * Ks[i]; }
            ph.push_back({w[0], Ms[0], Ks[0], e.v, zeros, e.ya});
            ph.push_back({w[1], Ms[1], Ks[1], e.ya, e.v, e.yb});
            ph.push_back({w[2], Ms[2], Ks[2], e.yb, zeros, e.yc});
            ph.push_back({w[3], Ms[3], Ks[3], e.yc, e.yb, e.v});
        }
    };
    const size_t FAT = 96 * 1024;
    auto gargs = [&](const ProbePhase& q) { GemvArgs a = {}; a.w[0] = q.w; a.M = q.M; a.K = q.K; a.x = q.x; a.resid = q.resid; a.y = q.y; return a; };
    auto K_a256 = k_gemv_sa<4, 2, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE>;
    auto K_a512 = k_gemv_sa<6, 1, 512, PRO_PLAIN, EPI_RESID, MAP_SINGLE>;
    auto K_b2 = k_gemv_sa<2, 2, 512, PRO_PLAIN, EPI_RESID, MAP_SINGLE>;
    auto K_b4 = k_gemv_sa<2, 4, 512, PRO_PLAIN, EPI_RESID, MAP_SINGLE>;
    for (auto k : {K_a256, K_a512, K_b2, K_b4}) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FAT));
    std::vector<ProbePhase> ph;
    std::vector<float> ref(d), got(d), refc(2 * F), gotc(2 * F);
    const double bytes = (double)per_layer * 4 * L;
    auto report = [&](const char* label, float ms) { printf("  %-64s %8.3f ms  %7.2f us/layer  %7.1f GB/s\n", label, ms, ms * 1e3 / L, bytes / ms / 1e6); fflush(stdout); };

    // ---- baselines on normal buffers, as a graph ----
    make_phases(exN, ph);
    auto run_graph = [&](int variant, const char* label) {   // 0 = product shapes, 1 = 512 threads U=2, 2 = 512 threads U=4
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (auto& q : ph) {
            GemvArgs a = gargs(q);
            if (q.K > 4096) hipLaunchKernelGGL(K_a512, dim3(nCU), dim3(512), FAT, st, a);
            else if (variant == 0) hipLaunchKernelGGL(K_a256, dim3(nCU), dim3(256), FAT, st, a);
            else if (variant == 1) hipLaunchKernelGGL(K_b2, dim3(nCU), dim3(512), FAT, st, a);
            else hipLaunchKernelGGL(K_b4, dim3(nCU), dim3(512), FAT, st, a);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms = time_ms([&] { CK(hipMemcpyAsync(exN.v, v0, d * 4, hipMemcpyDeviceToDevice, st)); CK(hipGraphLaunch(ge, st)); }, 5);
        report(label, ms);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    run_graph(0, "graph of product kernels (256 thr @K=4096, 512 @K=11008)");
    run_graph(1, "graph of k_gemv_sa, 512 thr, U=2 @K=4096");
    CK(hipMemcpy(ref.data(), exN.v, d * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(refc.data(), exN.yc, 2 * F * 4, hipMemcpyDeviceToHost));
    std::vector<float> ref2 = ref, refc2 = refc;
    run_graph(2, "graph of k_gemv_sa, 512 thr, U=4 @K=4096");
    std::vector<float> ref4(d), refc4(2 * F);
    CK(hipMemcpy(ref4.data(), exN.v, d * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(refc4.data(), exN.yc, 2 * F * 4, hipMemcpyDeviceToHost));
    {   double n2 = 0; for (float f : ref2) n2 += (double)f * f; printf("  |v| after %d layers = %.4f  (U=2 vs U=4 identical: %s)\n", L, sqrt(n2), memcmp(ref2.data(), ref4.data(), d * 4) ? "no" : "yes"); }

    // ---- persistent variants ----
    ProbePhase* dph; CK(hipMalloc(&dph, sizeof(ProbePhase) * L * 4));
    auto run_persist = [&](const Ex& e, int xm, int poll, int ud, int np2, int e2, int np6, int e6, uint32_t nowait = 0) {
        if (!e.ok) return;
        make_phases(e, ph);
        CK(hipMemcpy(dph, ph.data(), sizeof(ProbePhase) * ph.size(), hipMemcpyHostToDevice));
        PersistCtl c = {};
        c.count = e.count; c.err = e.err; c.arrivals_per_launch = (unsigned long long)(ph.size() - 1) * nCU; c.timeout_ticks = 300000; /* 3 ms */ c.dummy = dummy; c.nowait = nowait;
        const size_t lds = P_LDS_BYTES;
        bool found = false;
        auto launch = [&](void) {
            CK(hipMemcpyAsync(e.v, v0, d * 4, hipMemcpyDeviceToDevice, st));
#define LAUNCH(XM_, PL_, UD_, N2_, E2_, N6_, E6_) if (xm == XM_ && poll == PL_ && ud == UD_ && np2 == N2_ && e2 == E2_ && np6 == N6_ && e6 == E6_) { \
            auto k = k_persist_probe<XM_, PL_, UD_, N2_, E2_, N6_, E6_>; \
            CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); found = true; \
            hipLaunchKernelGGL(k, dim3(nCU), dim3(PTH), lds, st, (const ProbePhase*)dph, (uint32_t)ph.size(), c); }
            LAUNCH(XM_PLAIN, 0, 2, 0, 0, 0, 0) LAUNCH(XM_SCOPED, 0, 2, 0, 0, 0, 0)
            LAUNCH(XM_PLAIN, 0, 2, 8, 0, 3, 0) LAUNCH(XM_SCOPED, 0, 2, 8, 0, 3, 0) LAUNCH(XM_PLAIN, 1, 2, 8, 0, 3, 0)
            LAUNCH(XM_PLAIN, 0, 2, 4, 0, 2, 0) LAUNCH(XM_PLAIN, 0, 2, 10, 0, 3, 0)
            LAUNCH(XM_PLAIN, 0, 2, 8, 2, 3, 1) LAUNCH(XM_PLAIN, 0, 2, 8, 3, 3, 1) LAUNCH(XM_PLAIN, 0, 2, 8, 4, 3, 1) LAUNCH(XM_PLAIN, 0, 2, 8, 8, 3, 3)
            LAUNCH(XM_PLAIN, 0, 2, 4, 2, 2, 1) LAUNCH(XM_PLAIN, 0, 2, 4, 4, 2, 2) LAUNCH(XM_PLAIN, 0, 2, 2, 2, 1, 1) LAUNCH(XM_PLAIN, 0, 2, 10, 3, 3, 1) LAUNCH(XM_PLAIN, 0, 2, 10, 5, 3, 1)
            LAUNCH(XM_SCOPED, 0, 2, 8, 3, 3, 1) LAUNCH(XM_SCOPED, 0, 2, 4, 2, 2, 1)
            LAUNCH(XM_PLAIN, 0, 4, 8, 3, 3, 1) LAUNCH(XM_PLAIN, 0, 4, 8, 0, 3, 0)
            CK(hipGetLastError());
        };
        CK(hipMemset(e.err, 0, 4)); CK(hipMemset(e.count, 0, 8));
        float ms = time_ms(launch, 5);
        if (!found) { printf("  (variant not compiled: xm %d poll %d U %d np %d/%d e %d/%d)\n", xm, poll, ud, np2, np6, e2, e6); return; }
        uint32_t err; CK(hipMemcpy(&err, e.err, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(got.data(), e.v, d * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gotc.data(), e.yc, 2 * F * 4, hipMemcpyDeviceToHost));
        const std::vector<float>& r = ud == 2 ? ref2 : ref4; const std::vector<float>& rc = ud == 2 ? refc2 : refc4;
        const bool same = !memcmp(got.data(), r.data(), d * 4) && !memcmp(gotc.data(), rc.data(), 2 * F * 4);
        double md = 0; for (uint32_t i = 0; i < d; ++i) md = std::max(md, (double)fabsf(got[i] - r[i]));
        char label[160];
        snprintf(label, sizeof label, "persist buf=%-11s %s U=%d poll=%s park=%2d(%d early)/%d(%d) -> %s", e.name, xm == XM_SCOPED ? "scoped" : xm == XM_FENCE ? "fences" : "plain ", ud,
                 poll ? "scalar" : "vector", np2, e2, np6, e6, nowait ? "NOWAIT (ceiling)" : err ? "TIMEOUT" : same ? "bit-exact" : "MISMATCH");
        report(label, ms);
        if (!same && !err && !nowait) printf("      max |diff| %.3e\n", md);
    };
    if (!getenv("PROBE_SKIP_PERSIST")) {
    // ceiling of the resident stream: arrive, never wait
    run_persist(exU, XM_PLAIN, 0, 2, 0, 0, 0, 0, 1);
    run_persist(exU, XM_PLAIN, 0, 2, 8, 0, 3, 0, 1);
    run_persist(exU, XM_PLAIN, 0, 2, 8, 3, 3, 1, 1);
    run_persist(exU, XM_PLAIN, 0, 4, 8, 0, 3, 0, 1);
    // late parking only (first probe run), depth
    run_persist(exU, XM_PLAIN, 0, 2, 0, 0, 0, 0);
    run_persist(exU, XM_PLAIN, 0, 2, 4, 0, 2, 0);
    run_persist(exU, XM_PLAIN, 0, 2, 8, 0, 3, 0);
    run_persist(exU, XM_PLAIN, 0, 2, 10, 0, 3, 0);
    run_persist(exU, XM_PLAIN, 1, 2, 8, 0, 3, 0);
    run_persist(exN, XM_SCOPED, 0, 2, 8, 0, 3, 0);
    // early + late
    run_persist(exU, XM_PLAIN, 0, 2, 8, 2, 3, 1);
    run_persist(exU, XM_PLAIN, 0, 2, 8, 3, 3, 1);
    run_persist(exU, XM_PLAIN, 0, 2, 8, 4, 3, 1);
    run_persist(exU, XM_PLAIN, 0, 2, 8, 8, 3, 3);
    run_persist(exU, XM_PLAIN, 0, 2, 4, 2, 2, 1);
    run_persist(exU, XM_PLAIN, 0, 2, 4, 4, 2, 2);
    run_persist(exU, XM_PLAIN, 0, 2, 2, 2, 1, 1);
    run_persist(exU, XM_PLAIN, 0, 2, 10, 3, 3, 1);
    run_persist(exU, XM_PLAIN, 0, 2, 10, 5, 3, 1);
    run_persist(exN, XM_SCOPED, 0, 2, 8, 3, 3, 1);
    run_persist(exN, XM_SCOPED, 0, 2, 4, 2, 2, 1);
    run_persist(exU, XM_PLAIN, 0, 4, 8, 3, 3, 1);
    run_persist(exU, XM_PLAIN, 0, 4, 8, 0, 3, 0);
    }

    // ---- attention || wo in one launch vs two kernels (32 layers of wo only; the heads spin 3.0 us to land at ~4.9 us as a kernel) ----
    {
        const uint32_t H = 32;
        float *attn, *xres, *yo; unsigned int* sync;
        CK(hipMalloc(&attn, d * 4)); CK(hipMalloc(&xres, d * 4)); CK(hipMalloc(&yo, d * 4)); CK(hipMalloc(&sync, 256)); CK(hipMemset(sync, 0, 256));
        CK(hipMemcpy(xres, v0, d * 4, hipMemcpyDeviceToDevice));
        AwSync sy = {sync, sync + 32};
        PersistCtl c = {}; c.err = exN.err; c.count = exN.count; c.timeout_ticks = 300000; c.dummy = dummy;
        CK(hipMemset(exN.err, 0, 4));
        std::vector<const float*> wo; { size_t off = 0; for (int l = 0; l < L; ++l) { off += (size_t)Ms[0] * Ks[0]; wo.push_back(W + off); off += (size_t)Ms[1] * Ks[1] + (size_t)Ms[2] * Ks[2] + (size_t)Ms[3] * Ks[3]; } }
        auto ga = [&](int l) { GemvArgs a = {}; a.w[0] = wo[l]; a.M = d; a.K = d; a.x = attn; a.resid = xres; a.y = yo; return a; };
        for (uint32_t spin : {300u, 400u}) {
            // two kernels, as today (product shape: 256 threads for K = 4096)
            {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
                for (int l = 0; l < L; ++l) { hipLaunchKernelGGL(k_spin_heads, dim3(H), dim3(PTH), 0, st, spin, attn); hipLaunchKernelGGL(K_a256, dim3(nCU), dim3(256), FAT, st, ga(l)); }
                CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                float ms = time_ms([&] { CK(hipGraphLaunch(ge, st)); }, 5);
                printf("  heads(spin %u ticks) ; wo as two kernels                      %8.3f ms  %7.2f us/layer\n", spin, ms, ms * 1e3 / L);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
            {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
                for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k_spin_heads, dim3(H), dim3(PTH), 0, st, spin, attn);
                CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                float ms = time_ms([&] { CK(hipGraphLaunch(ge, st)); }, 5);
                printf("  heads(spin %u ticks) alone                                    %8.3f ms  %7.2f us/layer\n", spin, ms, ms * 1e3 / L);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
            }
#define AW(NP_, U_) { auto k = k_aw_probe<NP_, U_>; CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS_BYTES)); \
                hipGraph_t g; hipGraphExec_t ge; CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed)); \
                for (int l = 0; l < L; ++l) hipLaunchKernelGGL(k, dim3(H + nCU), dim3(PTH), P_LDS_BYTES, st, ga(l), c, sy, H, spin, attn); \
                CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); \
                float ms = time_ms([&] { CK(hipGraphLaunch(ge, st)); }, 5); \
                uint32_t err; CK(hipMemcpy(&err, exN.err, 4, hipMemcpyDeviceToHost)); unsigned hs[64]; CK(hipMemcpy(hs, sync, 256, hipMemcpyDeviceToHost)); \
                printf("  heads(spin %u) || wo in ONE launch, %2d rows parked, ring U=%d     %8.3f ms  %7.2f us/layer  err=%u cnt=%u done=%u\n", spin, NP_, U_, ms, ms * 1e3 / L, err, hs[0], hs[32]); \
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); }
            AW(16, 1) AW(12, 2) AW(8, 2) AW(4, 2) AW(0, 2)

            {   // forked capture: heads on a second stream, waiting wo on the first
                static hipStream_t s2 = nullptr; static hipEvent_t ef, ej;
                if (!s2) { CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming)); }
                auto k = k_wo_wait<12, 2>; const size_t lds = 72 * 1024;
                CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipGraph_t g; hipGraphExec_t ge; CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
                for (int l = 0; l < L; ++l) {
                    CK(hipEventRecord(ef, st)); CK(hipStreamWaitEvent(s2, ef, 0));
                    hipLaunchKernelGGL(k_heads_arrive, dim3(H), dim3(PTH), 0, s2, spin, attn, sy);
                    hipLaunchKernelGGL(k, dim3(nCU), dim3(PTH), lds, st, ga(l), c, sy, H);
                    CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(st, ej, 0));
                }
                CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                float ms = time_ms([&] { CK(hipGraphLaunch(ge, st)); }, 5);
                uint32_t err; CK(hipMemcpy(&err, exN.err, 4, hipMemcpyDeviceToHost)); unsigned hs[64]; CK(hipMemcpy(hs, sync, 256, hipMemcpyDeviceToHost));
                printf("  heads(spin %u) || wo as two launches, forked graph, 12 parked      %8.3f ms  %7.2f us/layer  err=%u cnt=%u done=%u\n", spin, ms, ms * 1e3 / L, err, hs[0], hs[32]);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipMemset(sync, 0, 256)); CK(hipMemset(exN.err, 0, 4));
            }
        }
        // correctness of the fused launch: same y as the two-kernel path with 512 threads
        {
            std::vector<float> y1(d), y2(d);
            hipLaunchKernelGGL(k_spin_heads, dim3(H), dim3(PTH), 0, st, 100u, attn); hipLaunchKernelGGL(K_b2, dim3(nCU), dim3(512), FAT, st, ga(3));
            CK(hipStreamSynchronize(st)); CK(hipMemcpy(y1.data(), yo, d * 4, hipMemcpyDeviceToHost)); CK(hipMemset(yo, 0, d * 4)); CK(hipMemset(attn, 0, d * 4));
            auto k = k_aw_probe<12, 2>; hipLaunchKernelGGL(k, dim3(H + nCU), dim3(PTH), P_LDS_BYTES, st, ga(3), c, sy, H, 100u, attn);
            CK(hipStreamSynchronize(st)); CK(hipMemcpy(y2.data(), yo, d * 4, hipMemcpyDeviceToHost));
            printf("  fused launch result vs two kernels: %s\n", memcmp(y1.data(), y2.data(), d * 4) ? "MISMATCH" : "bit-exact");
        }
    }
    printf("done\n");
    return 0;
}
