// tools/legacy_kernels.h — round-1 weight-stream kernels, superseded in the product by k_gemv_sa (csrc/kernels_llama.h) and k_gemv_q8s
// (csrc/kernels_q8.h) and removed from the library in round 3.  Kept ONLY so that tools/kernel_ablate.hip can still reproduce the
// ablation numbers in profiles/r01_kernel_ablation.txt and profiles/r02_q8s_ablation.txt.  Not built into libllamahip.so, not tested.
#pragma once
#include "../llama.go_amd/csrc/kernels_q8.h"

namespace lh {

template <int MAP>
__device__ __forceinline__ const f4* row_ptr(const GemvArgs& a, uint32_t v, uint32_t K4) {
    if (MAP == MAP_SINGLE) return (const f4*)a.w[0] + (size_t)v * K4;
    if (MAP == MAP_BLOCK) {
        const uint32_t m = v / a.rows_per_mat;
        return (const f4*)a.w[m] + (size_t)(v - m * a.rows_per_mat) * K4;
    }
    return (const f4*)a.w[v & 1] + (size_t)(v >> 1) * K4;
}

// RMSNorm + weight multiply on the thread's own columns (ml.go:1753-1812 then ml.go:1877-1914):
//   mean = (sum_f64 fl32(x*x)) / K ; scale = fl32(1/sqrt(mean + 1e-5)) ; t = fl32(x*scale) ; h = fl32(gamma*t)
template <int KI, int U, int TH, int PRO, int EPI, int MAP>
__global__ __launch_bounds__(TH) void k_gemv(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NW = TH / 64;
    double* sred = (double*)smem_raw;                // [NW]
    float* red = (float*)(smem_raw + NW * 8);        // [rows of this workgroup][NW] per-wave partial dot products
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t K4 = a.K >> 2;
    const uint32_t nwg = gridDim.x;
    // rows are dealt in pairs so RoPE / SiLU partners share a workgroup
    const uint32_t npairs = a.M >> 1;
    const uint32_t r0 = 2u * (uint32_t)(((uint64_t)blockIdx.x * npairs) / nwg);
    const uint32_t r1 = (blockIdx.x + 1 == nwg) ? a.M : 2u * (uint32_t)(((uint64_t)(blockIdx.x + 1) * npairs) / nwg);

    f4 xr[KI];
    f4 gr[KI];
    bool act[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        act[j] = (uint32_t)(tid + j * TH) < K4;
        xr[j] = act[j] ? ((const f4*)a.x)[tid + j * TH] : f4{0.f, 0.f, 0.f, 0.f};
        if (PRO == PRO_RMSNORM) gr[j] = act[j] ? ((const f4*)a.gamma)[tid + j * TH] : f4{0.f, 0.f, 0.f, 0.f};
    }
    // Epilogue operands of this workgroup's rows are fetched now (one finishing thread per row or row pair), so their
    // latency hides under the weight stream.
    const uint32_t fin = (EPI == EPI_STORE || EPI == EPI_RESID) ? (uint32_t)tid : 2u * (uint32_t)tid;  // row offset this thread finishes
    float resid_pre;
    double2 cs_pre;
    uint32_t past_pre;
    gemv_prefetch_fin<EPI>(a, r0, r1, fin, &resid_pre, &cs_pre, &past_pre);
    // first U rows are requested before the prologue so HBM latency overlaps the norm
    // Every load is UNCONDITIONAL: out-of-range rows / inactive lanes read a cache-resident dummy address instead of
    // branching.  Loads inside exec-masked branches make hipcc fall back to s_waitcnt vmcnt(0) right after the refills
    // (it cannot count them), which serialises the stream with the arithmetic.
    const f4* dummy = (const f4*)a.x;
    f4 w[U][KI];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool rv = r0 + u < r1;
        const f4* p = row_ptr<MAP>(a, rv ? r0 + u : r0, K4);
#pragma unroll
        for (int j = 0; j < KI; ++j) w[u][j] = ld_nt((rv && act[j]) ? p + tid + j * TH : dummy);
    }
    if (PRO == PRO_RMSNORM) rmsnorm_prologue<KI, TH>(xr, act, gr, a.K, sred);

    // Main stream: no workgroup barrier inside.  At the latency/bandwidth knee (U x 16 KiB in flight per CU) every stall
    // that delays the next load request costs throughput (tools/kernel_ablate: a per-batch barrier + epilogue = 3-4 %),
    // so waves run free and park their per-row partial sums in LDS.
    for (uint32_t r = r0; r < r1; r += U) {
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t nr = r + U + u;
            const bool nv = nr < r1;
            const f4* p = row_ptr<MAP>(a, nv ? nr : r0, K4);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const f4 c = w[u][j];
                s = fmaf(c.x, xr[j].x, s);
                s = fmaf(c.y, xr[j].y, s);
                s = fmaf(c.z, xr[j].z, s);
                s = fmaf(c.w, xr[j].w, s);
                w[u][j] = ld_nt((nv && act[j]) ? p + tid + j * TH : dummy);
            }
            acc[u] = s;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = wave_sum(acc[u]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (r + u < r1) red[(r - r0 + u) * NW + wave] = acc[u];
        }
    }
    __syncthreads();
    gemv_finish<EPI, NW>(a, red, r0, r1, fin, resid_pre, cs_pre, past_pre);
}

template <int MAP>
__device__ __forceinline__ void row_ptr_q8(const GemvArgs& a, uint32_t v, uint32_t K, const u4** q, const float** sc) {
    uint32_t m = 0, r = v;
    if (MAP == MAP_BLOCK) { m = v / a.rows_per_mat; r = v - m * a.rows_per_mat; }
    if (MAP == MAP_PAIR) { m = v & 1; r = v >> 1; }
    *q = (const u4*)((const signed char*)a.w[m] + (size_t)r * K);
    *sc = a.ws[m] + (size_t)r * (K >> 5);
}

// TPR threads share a row (TPR in {256, 1024}); G = 1024 / TPR rows are streamed side by side.
template <int KI, int U, int TPR, int PRO, int EPI, int MAP>
__global__ __launch_bounds__(1024) void k_gemv_q8(const GemvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TH = 1024, G = TH / TPR, NWR = TPR / 64;
    double* sred = (double*)smem_raw;            // [16]
    float* red = (float*)(smem_raw + 16 * 8);    // [rows of this workgroup][NWR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = tid / TPR, tr = tid % TPR, wr = wave % NWR;
    const uint32_t K = a.K, K16 = K >> 4;
    const uint32_t nwg = gridDim.x;
    const uint32_t npairs = a.M >> 1;
    const uint32_t r0 = 2u * (uint32_t)(((uint64_t)blockIdx.x * npairs) / nwg);
    const uint32_t r1 = (blockIdx.x + 1 == nwg) ? a.M : 2u * (uint32_t)(((uint64_t)(blockIdx.x + 1) * npairs) / nwg);

    f4 xr[KI][4];
    bool act[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        const uint32_t c = tr + j * TPR;
        act[j] = c < K16;
#pragma unroll
        for (int k = 0; k < 4; ++k) xr[j][k] = act[j] ? ((const f4*)a.x)[c * 4 + k] : f4{0.f, 0.f, 0.f, 0.f};
    }
    const uint32_t fin = (EPI == EPI_STORE || EPI == EPI_RESID) ? (uint32_t)tid : 2u * (uint32_t)tid;
    float resid_pre;
    double2 cs_pre;
    uint32_t past_pre;
    gemv_prefetch_fin<EPI>(a, r0, r1, fin, &resid_pre, &cs_pre, &past_pre);

    // slot u of this group holds row r0 + grp + G*(m + u).  Every load is unconditional (see k_gemv).
    u4 w[U][KI];
    float sc[U][KI];
    auto fetch = [&](u4 (&wd)[U][KI], float (&sd)[U][KI], uint32_t row_base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t row = row_base + G * u;
            const bool rv = row < r1;
            const u4* qp;
            const float* sp;
            row_ptr_q8<MAP>(a, rv ? row : r0, K, &qp, &sp);
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const uint32_t c = tr + j * TPR;
                const bool ok = rv && act[j];
                wd[u][j] = ld_nt_u4(ok ? qp + c : (const u4*)a.x);
                sd[u][j] = *(ok ? sp + (c >> 1) : a.x);
            }
        }
    };
    fetch(w, sc, r0 + grp);

    if (PRO == PRO_RMSNORM) {
        // RMSNorm * gamma on the thread's own 16*KI columns; every row group holds the same x and reduces it identically
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < KI; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (act[j]) {
                    s += (double)__fmul_rn(xr[j][k].x, xr[j][k].x);
                    s += (double)__fmul_rn(xr[j][k].y, xr[j][k].y);
                    s += (double)__fmul_rn(xr[j][k].z, xr[j][k].z);
                    s += (double)__fmul_rn(xr[j][k].w, xr[j][k].w);
                }
            }
        s = wave_sum_f64(s);
        if (lane == 0) sred[wave] = s;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < NWR; ++k) tot += sred[grp * NWR + k];
        const float scale = (float)(1.0 / sqrt(tot / (double)K + 1e-5));
#pragma unroll
        for (int j = 0; j < KI; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (act[j]) {
                    const f4 g = ((const f4*)a.gamma)[(tr + j * TPR) * 4 + k];
                    xr[j][k].x = __fmul_rn(g.x, __fmul_rn(xr[j][k].x, scale));
                    xr[j][k].y = __fmul_rn(g.y, __fmul_rn(xr[j][k].y, scale));
                    xr[j][k].z = __fmul_rn(g.z, __fmul_rn(xr[j][k].z, scale));
                    xr[j][k].w = __fmul_rn(g.w, __fmul_rn(xr[j][k].w, scale));
                }
            }
    }

    // The arithmetic per byte is 12x the fp32 kernel's (16 cvt + 16 fma per 16-byte load), so the next batch is requested
    // into a SECOND register set at the top of the iteration: with in-place refills hipcc sinks the loads below the dot
    // products (WAR on the slot registers) and the wave then idles a full memory latency per batch.
    for (uint32_t rb = r0 + grp; rb < r1; rb += G * U) {
        u4 wn[U][KI];
        float scn[U][KI];
        fetch(wn, scn, rb + G * U);
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KI; ++j) s = fmaf(sc[u][j], dot16_q8(w[u][j], xr[j]), s);
            acc[u] = s;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = wave_sum_lane63(acc[u]);
        if (lane == 63) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t row = rb + G * u;
                if (row < r1) red[(row - r0) * NWR + wr] = acc[u];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < KI; ++j) { w[u][j] = wn[u][j]; sc[u][j] = scn[u][j]; }
    }
    __syncthreads();
    gemv_finish<EPI, NWR>(a, red, r0, r1, fin, resid_pre, cs_pre, past_pre);
}


}  // namespace lh
