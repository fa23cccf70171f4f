// csrc/kernels_llama.h — fused gfx950 kernels of the LLaMA plan (decode N=1 and small-N prefill).
//
// Replaces, per SURVEY §8a: rows 6 (MulMat weights), 8 (GetRows), 9-11 (RMSNorm/Repeat/Mul), 12 (Cpy into
// cache / merges), 13 (Rope), 15-18 (Scale/DiagMaskInf/SoftMax/KQ/KQV), 19 (Silu), 20 (Add).
//
// GEMV layout ("fat workgroup", chosen from profiles/r01_gemv_probe.txt: 6.4-6.7 TB/s on 7B shapes):
//   grid = #CU workgroups of TH = 1024 threads (16 waves); a dynamic-LDS request > 80 KiB keeps a second
//   workgroup off the CU, so every CU owns one contiguous, equally sized block of weight rows and the whole
//   chip finishes together.  Thread t owns columns 4t..4t+3 (+ 4*TH*j): the activation vector lives in KI
//   float4 REGISTERS for the whole kernel (no LDS staging, no re-reads); weight rows stream through U
//   rotating float4 slots per thread (non-temporal global_load_dwordx4: 1 KiB contiguous per wave-instruction,
//   U x 16 KiB in flight per CU).  Per row: 4*KI FMAs, a DPP wave reduction, one LDS word per wave, and a
//   fixed-order cross-wave sum -> bit-reproducible run to run.
#pragma once
#include "kernels_common.h"

namespace lh {

enum { PRO_PLAIN = 0, PRO_RMSNORM = 1 };
enum { EPI_STORE = 0, EPI_RESID = 1, EPI_QKV_ROPE = 2, EPI_SILU_MUL = 3 };
enum { MAP_SINGLE = 0, MAP_BLOCK = 1, MAP_PAIR = 2 };

struct GemvArgs {
    const float* w[3];      // matrix bases (MAP_BLOCK: [wq,wk,wv]; MAP_PAIR: [w1,w3]); block-int8: the int8 planes
    const float* ws[3];     // block-int8 only: per-32-column scales [rows][K/32]
    uint32_t rows_per_mat;  // MAP_BLOCK
    uint32_t M;             // virtual rows
    uint32_t K;             // columns
    const float* x;         // activation [K]
    const float* gamma;     // PRO_RMSNORM: norm weight [K]
    float* y;               // EPI_STORE / EPI_RESID / EPI_SILU_MUL output
    const float* resid;     // EPI_RESID
    float* q_out;           // EPI_QKV_ROPE: roped Q [d]
    float* k_cache;         // EPI_QKV_ROPE: this layer's K slot base [ctx][d]
    float* v_cache;
    const double2* rope;    // [pos][hd/2] (cos, sin)
    uint32_t hd;            // head dim (= rope dims)
    uint32_t d;             // embd
    const StepParams* sp;   // past
    uint32_t wg_q, wg_r;    // (M / 2) / #workgroups and its remainder (wg_row_block; filled by launch_gemv / launch_gemv_q8)
};

template <int KI, int TH>
__device__ __forceinline__ void rmsnorm_prologue(f4 (&xr)[KI], const bool (&act)[KI], const f4 (&gr)[KI], uint32_t K, double* sred) {
    constexpr int NW = TH / 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        if (act[j]) {
            s += (double)__fmul_rn(xr[j].x, xr[j].x);
            s += (double)__fmul_rn(xr[j].y, xr[j].y);
            s += (double)__fmul_rn(xr[j].z, xr[j].z);
            s += (double)__fmul_rn(xr[j].w, xr[j].w);
        }
    }
    s = wave_sum_f64(s);
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += sred[w];
    const double mean = tot / (double)K;
    const float scale = (float)(1.0 / sqrt(mean + 1e-5));
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        if (act[j]) {
            const f4 g = gr[j];
            xr[j].x = __fmul_rn(g.x, __fmul_rn(xr[j].x, scale));
            xr[j].y = __fmul_rn(g.y, __fmul_rn(xr[j].y, scale));
            xr[j].z = __fmul_rn(g.z, __fmul_rn(xr[j].z, scale));
            xr[j].w = __fmul_rn(g.w, __fmul_rn(xr[j].w, scale));
        }
    }
}

// Epilogue, once per workgroup, all rows in parallel; cross-wave sums in fixed order -> bit-reproducible.
// NP = partial sums per row (waves that shared the row).
template <int EPI, int NP>
__device__ __forceinline__ void gemv_finish(const GemvArgs& a, const float* red, uint32_t r0, uint32_t r1, uint32_t fin, float resid_pre,
                                            double2 cs_pre, uint32_t past_pre) {
    if (r0 + fin >= r1) return;
    const float* p0 = red + fin * NP;
    float s0 = 0.f;
#pragma unroll
    for (int k = 0; k < NP; ++k) s0 += p0[k];
    const uint32_t v = r0 + fin;
    if (EPI == EPI_STORE) {
        a.y[v] = s0;
    } else if (EPI == EPI_RESID) {
        a.y[v] = __fadd_rn(s0, resid_pre);  // Add(cur, inp) ml.go:2515-2584
    } else {
        const float* p1 = p0 + NP;
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < NP; ++k) s1 += p1[k];
        if (EPI == EPI_SILU_MUL) {
            // Silu(w1 h) then Mul(., w3 h): ml.go:2587-2589, 1877-1914 (llama.go:354-361)
            a.y[v >> 1] = __fmul_rn(silu_ref(s0), s1);
        } else {  // EPI_QKV_ROPE: Rope mode 0 on Q / mode 1 on the new K row (ml.go:2253-2328), K,V appended to the cache (llama.go:274-278)
            const uint32_t d = a.d;
            if (v < 2 * d) {
                const uint32_t e = v < d ? v : v - d;
                float o0, o1;
                rope_rotate(s0, s1, cs_pre, &o0, &o1);
                float* dst = v < d ? a.q_out + e : a.k_cache + (size_t)past_pre * d + e;
                dst[0] = o0;
                dst[1] = o1;
            } else {
                float* dst = a.v_cache + (size_t)past_pre * d + (v - 2 * d);
                dst[0] = s0;
                dst[1] = s1;
            }
        }
    }
}

// Epilogue operands of the row (pair) this thread will finish, fetched at kernel start so their latency hides under the
// weight stream.
template <int EPI>
__device__ __forceinline__ void gemv_prefetch_fin(const GemvArgs& a, uint32_t r0, uint32_t r1, uint32_t fin, float* resid_pre, double2* cs_pre,
                                                  uint32_t* past_pre) {
    *resid_pre = 0.f;
    *cs_pre = double2{1.0, 0.0};
    *past_pre = 0;
    if (EPI == EPI_RESID) {
        if (r0 + fin < r1) *resid_pre = a.resid[r0 + fin];
    } else if (EPI == EPI_QKV_ROPE) {
        *past_pre = a.sp->past;
        const uint32_t v = r0 + fin;
        if (v < r1 && v < 2 * a.d) {
            const uint32_t e = v < a.d ? v : v - a.d;
            *cs_pre = a.rope[(size_t)*past_pre * (a.hd >> 1) + ((e % a.hd) >> 1)];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// k_gemv_sa: the fp32 weight stream with SCALAR row addressing.  In its predecessor (k_gemv, round 1; kept in tools/legacy_kernels.h for
// the ablation probe) every weight load takes a per-lane 64-bit pointer that is selected
// (v_cndmask x2), offset (v_lshl_add_u64) and, for grouped matrices, rebuilt from a pointer fetched out of the kernel arguments
// per row: ~24 VALU instructions per row pair next to 32 FMAs, and the compiler parks all refills behind them at the end of
// the iteration.  Here the row base is a wave-uniform (SGPR) pointer — matrix bases hoisted into registers once, row index
// scalar, out-of-range rows redirected to the cache-resident activation vector by a scalar select — and the lane contributes a
// constant 32-bit byte offset: a load is `global_load_dwordx4 v, v_off, s[base]` with no vector address arithmetic at all.
// Same arithmetic, same summation order, same results bit for bit as k_gemv.
// ---------------------------------------------------------------------------------------------------
template <int MAP>
__device__ __forceinline__ const char* gemv_row_base(const char* w0, const char* w1, const char* w2, uint32_t rows_per_mat, uint32_t v, uint64_t row_bytes) {
    if (MAP == MAP_SINGLE) return w0 + (uint64_t)v * row_bytes;
    if (MAP == MAP_BLOCK) {
        const uint32_t m = (v >= rows_per_mat ? 1u : 0u) + (v >= 2u * rows_per_mat ? 1u : 0u);
        const char* b = m == 0 ? w0 : (m == 1 ? w1 : w2);
        return b + (uint64_t)(v - m * rows_per_mat) * row_bytes;
    }
    return ((v & 1u) ? w1 : w0) + (uint64_t)(v >> 1) * row_bytes;
}

template <int KI, int U, int TH, int PRO, int EPI, int MAP>
__global__ __launch_bounds__(TH) void k_gemv_sa(const GemvArgs a) {
    LH_TOUCH_ARGS(a.w[0], a.x, a.hd, a.wg_r);   // the argument block's lines (0x00 / 0x40 / 0x80 / 0x94) behind one wait
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NW = TH / 64;
    double* sred = (double*)smem_raw;                // [NW]
    float* red = (float*)(smem_raw + NW * 8);        // [rows of this workgroup][NW] per-wave partial dot products
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t K4 = a.K >> 2;
    uint32_t r0, r1;
    wg_row_block(a.M, a.wg_q, a.wg_r, &r0, &r1);
    const char *w0 = (const char*)a.w[0], *w1 = (const char*)a.w[1], *w2 = (const char*)a.w[2];
    const char* xdummy = (const char*)a.x;           // K floats = exactly one row's extent: any lane offset stays inside it
    const uint32_t rpm = a.rows_per_mat;
    const uint64_t row_bytes = (uint64_t)a.K * 4;

    // (epilogue operands first: the position -> RoPE entry chain of wq|wk|wv flies while the x / gamma loads are issued - see k_gemv_q8s)
    const uint32_t fin = (EPI == EPI_STORE || EPI == EPI_RESID) ? (uint32_t)tid : 2u * (uint32_t)tid;
    float resid_pre;
    double2 cs_pre;
    uint32_t past_pre;
    gemv_prefetch_fin<EPI>(a, r0, r1, fin, &resid_pre, &cs_pre, &past_pre);
    f4 xr[KI];
    f4 gr[KI];
    bool act[KI];
    uint32_t loff[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        act[j] = (uint32_t)(tid + j * TH) < K4;
        loff[j] = act[j] ? (uint32_t)(tid + j * TH) * 16u : 0u;
        xr[j] = act[j] ? ((const f4*)a.x)[tid + j * TH] : f4{0.f, 0.f, 0.f, 0.f};
        if (PRO == PRO_RMSNORM) gr[j] = act[j] ? ((const f4*)a.gamma)[tid + j * TH] : f4{0.f, 0.f, 0.f, 0.f};
    }
    f4 w[U][KI];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const char* p = (r0 + u < r1) ? gemv_row_base<MAP>(w0, w1, w2, rpm, r0 + u, row_bytes) : xdummy;
#pragma unroll
        for (int j = 0; j < KI; ++j) w[u][j] = ld_nt((const f4*)(p + loff[j]));
    }
    if (PRO == PRO_RMSNORM) rmsnorm_prologue<KI, TH>(xr, act, gr, a.K, sred);
    // inactive lanes (K not a multiple of 4*TH) multiply whatever they loaded by x = 0
    for (uint32_t r = r0; r < r1; r += U) {
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t nr = r + U + u;
            const char* p = nr < r1 ? gemv_row_base<MAP>(w0, w1, w2, rpm, nr, row_bytes) : xdummy;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const f4 c = w[u][j];
                s = fmaf(c.x, xr[j].x, s);
                s = fmaf(c.y, xr[j].y, s);
                s = fmaf(c.z, xr[j].z, s);
                s = fmaf(c.w, xr[j].w, s);
                w[u][j] = ld_nt((const f4*)(p + loff[j]));
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

// ---------------------------------------------------------------------------------------------------
// Multi-column variant for small-N prefill (N <= NC per launch): the weight row is still streamed
// once, NC activation columns live in registers.  Y[c][v] (+ resid) row-major [N][M].
// ---------------------------------------------------------------------------------------------------
struct GemmColsArgs {
    const float* w;     // [M][K]
    const float* x;     // [ncols][K] (row c at x + c*ldx)
    float* y;           // [ncols][M] (row c at y + c*ldy)
    const float* resid; // optional, same layout as y
    uint32_t M, K, ldx, ldy, ncols;
};

template <int KI, int U, int TH, int NC>
__global__ __launch_bounds__(TH) void k_gemv_cols(const GemmColsArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NW = TH / 64;
    float* red = (float*)smem_raw;  // [2][U*NC][NW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t K4 = a.K >> 2;
    const uint32_t nwg = gridDim.x;
    const uint32_t r0 = (uint32_t)(((uint64_t)blockIdx.x * a.M) / nwg);
    const uint32_t r1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * a.M) / nwg);
    f4 xr[NC][KI];
    bool act[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) act[j] = (uint32_t)(tid + j * TH) < K4;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < KI; ++j)
            xr[c][j] = (act[j] && (uint32_t)c < a.ncols) ? ((const f4*)(a.x + (size_t)c * a.ldx))[tid + j * TH] : f4{0.f, 0.f, 0.f, 0.f};
    const f4* dummy = (const f4*)a.x;  // unconditional loads (see k_gemv)
    f4 w[U][KI];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const bool rv = r0 + u < r1;
        const f4* p = (const f4*)a.w + (size_t)(rv ? r0 + u : r0) * K4;
#pragma unroll
        for (int j = 0; j < KI; ++j) w[u][j] = ld_nt((rv && act[j]) ? p + tid + j * TH : dummy);
    }
    int buf = 0;
    for (uint32_t r = r0; r < r1; r += U) {
        float acc[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t nr = r + U + u;
            const bool nv = nr < r1;
            const f4* p = (const f4*)a.w + (size_t)(nv ? nr : r0) * K4;
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[u][c] = 0.f;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const f4 cw = w[u][j];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float s = acc[u][c];
                    s = fmaf(cw.x, xr[c][j].x, s);
                    s = fmaf(cw.y, xr[c][j].y, s);
                    s = fmaf(cw.z, xr[c][j].z, s);
                    s = fmaf(cw.w, xr[c][j].w, s);
                    acc[u][c] = s;
                }
                w[u][j] = ld_nt((nv && act[j]) ? p + tid + j * TH : dummy);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float s = wave_sum_lane63(acc[u][c]);  // 6 DPP adds, total in lane 63 (the U*NC reductions outweigh the FMAs here)
                if (lane == 63) red[((buf * U + u) * NC + c) * NW + wave] = s;
            }
        __syncthreads();
        if (tid < U * NC) {
            const int u = tid / NC, c = tid % NC;
            if (r + u < r1 && (uint32_t)c < a.ncols) {
                const float* p = red + ((buf * U + u) * NC + c) * NW;
                float s = 0.f;
#pragma unroll
                for (int k = 0; k < NW; ++k) s += p[k];
                const size_t o = (size_t)c * a.ldy + (r + u);
                if (a.resid) s = __fadd_rn(s, a.resid[o]);
                a.y[o] = s;
            }
        }
        buf ^= 1;
    }
}

// ---------------------------------------------------------------------------------------------------
// Attention for one (head, query row): scores over the cached keys, scale, causal mask, softmax, PV.
// Replaces KQ / Scale / DiagMaskInf / SoftMax / VTrans copy / KQV / merge (llama.go:300-333) without
// materialising the transposed V or the [T x N x H] score tensor.  Keys beyond the causal limit are
// skipped: in the reference they become exactly 0 after the softmax (ml.go:2476-2477) and add nothing.
//   grid = (H, N), 1024 threads.  K/V rows are strided by d floats in the cache; 32 groups of 32 lanes each take one key
//   (float4 per lane, 128-float head) with 4 keys in flight per group, reduce with DPP.  Scores of the row live in LDS.
// ---------------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* q;        // [N][d] roped queries
    const float* k_cache;  // layer slot base [ctx][d] (post-RoPE keys)
    const float* v_cache;
    float* out;            // [N][d] merged heads
    uint32_t d, hd, n;
    float scale;           // fl32(1/sqrt(hd)) llama.go:306
    const StepParams* sp;  // past (device) ...
    uint32_t past_host;    // ... or host value when sp == nullptr
    // batched Eval (lh_batch): query row j belongs to its own stream: keys 0..rows[j].pos of the cache rows[j].kc / .vc (+ kv_off floats: the
    // layer's slot); k_cache, v_cache, sp and past_host are then unused
    const BatchRow* rows;
    uint64_t kv_off;
    // block-int8 on the bf16 pipe (k_stream_q8b): the merged heads ALSO as three bf16 planes (out = hi + mid + lo exactly), row j at
    // out_s3 + p * out_plane + j * d - the activation format of the wo launch behind it
    uint16_t* out_s3;
    uint64_t out_plane;
};
__device__ __forceinline__ void attn_store_split3(const AttnArgs& a, size_t idx, float o) {
    const uint32_t h = __builtin_bit_cast(uint32_t, o) & 0xffff0000u;
    const float r1 = __fsub_rn(o, __builtin_bit_cast(float, h));
    const uint32_t m = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
    const float r2 = __fsub_rn(r1, __builtin_bit_cast(float, m));
    a.out_s3[idx] = (uint16_t)(h >> 16); a.out_s3[idx + a.out_plane] = (uint16_t)(m >> 16); a.out_s3[idx + 2 * a.out_plane] = (uint16_t)(__builtin_bit_cast(uint32_t, r2) >> 16);
}

constexpr int ATT_TH = 1024;

__global__ __launch_bounds__(ATT_TH) void k_attention(const AttnArgs a) {
    LH_TOUCH_ARGS(a.q, a.sp, a.rows);   // both lines of the argument block at once (rows -> sp -> position was three dependent scalar misses)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NWV = ATT_TH / 64, NG = ATT_TH / 32;  // waves, 32-lane key groups
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t h = blockIdx.x, j = blockIdx.y;
    const uint32_t past = a.rows ? a.rows[j].pos : (a.sp ? a.sp->past : a.past_host);
    const uint32_t T = a.rows ? past + 1 : past + j + 1;  // keys 0..past+j are visible to query j (mask: i > past + j, ml.go:2401-2404)
    const uint32_t Tp = (T + 63) & ~63u;
    float* sc = (float*)smem_raw;     // [Tp] scaled scores
    float* pr = sc + Tp;              // [Tp] un-normalised probabilities
    float* scratch = pr + Tp;         // [ATT_TH] PV partials / reduction scratch
    const uint32_t d = a.d, hd = a.hd;
    const float* q = a.q + (size_t)j * d + h * hd;
    // (the cache pointers come out of a select between a kernel argument and a pointer read from the row table: say that they are GLOBAL memory, or every
    // K / V load is a flat_load - counted on the LDS counter too, so each wait for cache rows also drained the LDS queue; round 6, ISA of this kernel)
    typedef const float __attribute__((address_space(1))) gfl;
    typedef const f4 __attribute__((address_space(1))) gf4;
    gfl* Kc = (gfl*)(uintptr_t)((a.rows ? a.rows[j].kc + a.kv_off : a.k_cache) + h * hd);
    gfl* Vc = (gfl*)(uintptr_t)((a.rows ? a.rows[j].vc + a.kv_off : a.v_cache) + h * hd);
    // The cache rows of one head are 512 B segments strided by embd: every loop below keeps several INDEPENDENT row
    // loads in flight per lane (a dependent one-row-per-iteration loop costs a full L2 latency per key: 0.27 us/key measured).
    const uint32_t phases = ATT_TH / hd;  // hd = 128 -> 8 key phases in the PV step
    const uint32_t c = tid % hd, ph = tid / hd;
    constexpr int VP = 8;
    float vpre[VP];
#pragma unroll
    for (int i = 0; i < VP; ++i) {   // first V rows: issued before anything else, consumed last
        const uint32_t t = ph + (uint32_t)i * phases;
        vpre[i] = t < T ? Vc[(size_t)t * d + c] : 0.f;
    }
    // --- scores: one key per 32-lane group, UN keys in flight per group (hd = 128 -> float4 per lane; other hd: strided loop)
    const int g = tid >> 5, gl = tid & 31;
    constexpr int UN = 4;
    if (hd == 128) {
        const f4 qv = *(const f4*)(q + gl * 4);
        for (uint32_t t0 = g; t0 < T; t0 += NG * UN) {
            f4 kv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t t = t0 + u * NG;
                kv[u] = *(gf4*)(Kc + (size_t)(t < T ? t : 0) * d + gl * 4);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t t = t0 + u * NG;
                float s = fmaf(kv[u].x, qv.x, 0.f);
                s = fmaf(kv[u].y, qv.y, s); s = fmaf(kv[u].z, qv.z, s); s = fmaf(kv[u].w, qv.w, s);
                s = half_wave_sum(s);
                if (gl == 0 && t < T) sc[t] = __fmul_rn(s, a.scale);  // Scale ml.go:2331-2374
            }
        }
    } else {
        for (uint32_t t = g; t < T; t += NG) {
            float s = 0.f;
            for (uint32_t cc = gl * 4; cc < hd; cc += 128) {
                const f4 kv = *(gf4*)(Kc + (size_t)t * d + cc);
                const f4 qv = *(const f4*)(q + cc);
                s = fmaf(kv.x, qv.x, s); s = fmaf(kv.y, qv.y, s); s = fmaf(kv.z, qv.z, s); s = fmaf(kv.w, qv.w, s);
            }
            s = half_wave_sum(s);
            if (gl == 0) sc[t] = __fmul_rn(s, a.scale);
        }
    }
    __syncthreads();
    // --- softmax (ml.go:2432-2505): max, p = fl32(exp_f64(fl32(s - max))), fp32 sum, p *= 1/sum
    float inv;
    if (T <= 64) {
        // rows of at most 64 keys (one per lane): wave 0 alone takes the maximum, the exponentials, the sum and 1 / sum and leaves the latter where the scores
        // were; the other fifteen waves wait at the barrier instead of issuing the same reductions four deep per SIMD (the contention showed as ~1000 clocks
        // of waiting at the second barrier: s_memtime stamps, tools/attn_latency_probe).  The same operations in the same order as the path below: same bits.
        if (wave == 0) {
            float m = (uint32_t)lane < T ? sc[lane] : -INFINITY;
            m = wave_max(m);
            float psum = 0.f;
            if ((uint32_t)lane < T) {
                const float p = (float)exp((double)__fsub_rn(sc[lane], m));
                pr[lane] = p;
                psum += p;
            }
            psum = wave_sum(psum);
            if (lane == 0) sc[0] = __fdiv_rn(1.0f, psum);   // (every lane of this wave has read its score above; nobody reads the scores again)
        }
        __syncthreads();
        inv = sc[0];
    } else if (T <= 128) {
        // short rows: every wave takes the row's maximum and sum with wave-level reductions (same code -> same bits)
        float m = -INFINITY;
        for (uint32_t t = lane; t < T; t += 64) m = fmaxf(m, sc[t]);
        m = wave_max(m);
        // The f64 exponentials ONCE per row, not once per wave: waves 0 / 1 take keys 0..63 / 64..127 (one per lane).  Until round 6 every one of the
        // sixteen waves evaluated them - the same instruction stream four times per SIMD, ~1000 clocks of issue contention that the workgroup's second
        // barrier then waited out (s_memtime stamps, tools/attn_latency_probe).  The sums below keep their order: same bits.
        if (wave < 2) {
            const uint32_t t = (uint32_t)lane + 64u * (uint32_t)wave;
            if (t < T) pr[t] = (float)exp((double)__fsub_rn(sc[t], m));
        }
        __syncthreads();
        float psum = 0.f;
        for (uint32_t t = lane; t < T; t += 64) psum += pr[t];
        psum = wave_sum(psum);
        inv = __fdiv_rn(1.0f, psum);
    } else {
        // long rows: the f64 exps are spread over all threads, two block reductions in fixed order
        float m = -INFINITY;
        for (uint32_t t = tid; t < T; t += ATT_TH) m = fmaxf(m, sc[t]);
        m = wave_max(m);
        if (lane == 0) scratch[wave] = m;
        __syncthreads();
        m = scratch[0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) m = fmaxf(m, scratch[w]);
        __syncthreads();
        float psum = 0.f;
        for (uint32_t t = tid; t < T; t += ATT_TH) {
            const float p = (float)exp((double)__fsub_rn(sc[t], m));
            pr[t] = p;
            psum += p;
        }
        psum = wave_sum(psum);
        if (lane == 0) scratch[wave] = psum;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) tot += scratch[w];
        inv = __fdiv_rn(1.0f, tot);
        __syncthreads();
    }
    // --- PV: thread (c, ph) accumulates its key phase, VP independent row loads in flight (pr[] is complete: barrier above on both paths)
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < VP; ++i) {
        const uint32_t t = ph + (uint32_t)i * phases;
        if (t < T) acc = fmaf(vpre[i], __fmul_rn(pr[t], inv), acc);
    }
    for (uint32_t t0 = ph + VP * phases; t0 < T; t0 += VP * phases) {
        float vv[VP];
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const uint32_t t = t0 + (uint32_t)i * phases;
            vv[i] = Vc[(size_t)(t < T ? t : 0) * d + c];
        }
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const uint32_t t = t0 + (uint32_t)i * phases;
            if (t < T) acc = fmaf(vv[i], __fmul_rn(pr[t], inv), acc);
        }
    }
    scratch[tid] = acc;
    __syncthreads();
    if (tid < (int)hd) {
        float o = scratch[tid];
        for (uint32_t p2 = 1; p2 < phases; ++p2) o += scratch[tid + p2 * hd];
        a.out[(size_t)j * d + h * hd + tid] = o;
        if (a.out_s3) attn_store_split3(a, (size_t)j * d + h * hd + tid, o);
    }
}

// ---------------------------------------------------------------------------------------------------
// Long contexts (plans with ctx > 256), decode: one workgroup per head reads 2*T*hd*4 bytes alone (1 MB at T = 1024) and
// 32 workgroups cannot pull that out of L2 fast enough (38 us per layer at T = 1000).  Split the keys of a head into
// chunks of ATT_TC, one workgroup per (head, chunk) writes an un-normalised partial {o_c[hd], m_c, l_c} (softmax local to
// the chunk), and k_attention_combine merges the chunks: M = max m_c, w_c = exp(m_c - M), l = sum l_c w_c,
// out = (sum o_c w_c) / l.  Same mathematics as the single pass, two more fp32 roundings per output (<= 1e-7 relative).
// The grid is static (ceil(ctx / ATT_TC) chunks, so one captured hipGraph serves every position); chunks beyond the
// current length exit immediately.
// ---------------------------------------------------------------------------------------------------
constexpr int ATT_TC = 128;

__global__ __launch_bounds__(ATT_TH) void k_attention_split(const AttnArgs a, float* __restrict__ part) {
    LH_TOUCH_ARGS(a.q, a.sp, a.rows);
    __shared__ float sc[ATT_TC];
    __shared__ float pr[ATT_TC];
    __shared__ float scratch[ATT_TH];
    constexpr int NG = ATT_TH / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t h = blockIdx.x, ch = blockIdx.y, nch = gridDim.y, j = blockIdx.z;   // j: row of a batched Eval (0 otherwise)
    const uint32_t past = a.rows ? a.rows[j].pos : (a.sp ? a.sp->past : a.past_host);
    const uint32_t T = past + 1, c0 = ch * ATT_TC;
    if (c0 >= T) return;
    const uint32_t Tl = (T - c0 < (uint32_t)ATT_TC) ? T - c0 : (uint32_t)ATT_TC;  // keys of this chunk
    const uint32_t d = a.d, hd = a.hd;
    const float* q = a.q + (size_t)j * d + h * hd;
    typedef const float __attribute__((address_space(1))) gfl;   // (global, not flat: see k_attention)
    typedef const f4 __attribute__((address_space(1))) gf4;
    gfl* Kc = (gfl*)(uintptr_t)((a.rows ? a.rows[j].kc + a.kv_off : a.k_cache) + (size_t)c0 * d + h * hd);
    gfl* Vc = (gfl*)(uintptr_t)((a.rows ? a.rows[j].vc + a.kv_off : a.v_cache) + (size_t)c0 * d + h * hd);
    const uint32_t phases = ATT_TH / hd, c = tid % hd, ph = tid / hd;
    constexpr int VP = ATT_TC / 8;  // hd = 128: 8 phases x 16 keys = the whole chunk in flight
    float vpre[VP];
#pragma unroll
    for (int i = 0; i < VP; ++i) {
        const uint32_t t = ph + (uint32_t)i * phases;
        vpre[i] = t < Tl ? Vc[(size_t)t * d + c] : 0.f;
    }
    const int g = tid >> 5, gl = tid & 31;
    {   // scores of the chunk in one batch: 32 groups x 4 keys, all K rows requested before the first is used (hd = 128)
        constexpr int UN = ATT_TC / NG;
        const f4 qv = *(const f4*)(q + gl * 4);
        f4 kv[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t t = g + u * NG;
            kv[u] = *(gf4*)(Kc + (size_t)(t < Tl ? t : 0) * d + gl * 4);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const uint32_t t = g + u * NG;
            float s = fmaf(kv[u].x, qv.x, 0.f);
            s = fmaf(kv[u].y, qv.y, s); s = fmaf(kv[u].z, qv.z, s); s = fmaf(kv[u].w, qv.w, s);
            s = half_wave_sum(s);
            if (gl == 0 && t < Tl) sc[t] = __fmul_rn(s, a.scale);  // Scale ml.go:2331-2374
        }
    }
    __syncthreads();
    // chunk-local softmax statistics: maximum and sum per wave (same code -> same bits), the f64 exponentials once per chunk (waves 0 / 1 take keys
    // 0..63 / 64..127; sixteen waves evaluating all of them was the same instruction stream four times per SIMD - see k_attention)
    float m = -INFINITY;
    for (uint32_t t = lane; t < Tl; t += 64) m = fmaxf(m, sc[t]);
    m = wave_max(m);
    {
        const int wave = tid >> 6;
        if (wave < 2) {
            const uint32_t t = (uint32_t)lane + 64u * (uint32_t)wave;
            if (t < Tl) pr[t] = (float)exp((double)__fsub_rn(sc[t], m));
        }
    }
    __syncthreads();
    float psum = 0.f;
    for (uint32_t t = lane; t < Tl; t += 64) psum += pr[t];
    psum = wave_sum(psum);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < VP; ++i) {
        const uint32_t t = ph + (uint32_t)i * phases;
        if (t < Tl) acc = fmaf(vpre[i], pr[t], acc);
    }
    scratch[tid] = acc;
    __syncthreads();
    float* dst = part + (((size_t)j * gridDim.x + h) * nch + ch) * (hd + 2);
    if (tid < (int)hd) {
        float o = scratch[tid];
        for (uint32_t p2 = 1; p2 < phases; ++p2) o += scratch[tid + p2 * hd];
        dst[tid] = o;
    }
    if (tid == 0) { dst[hd] = m; dst[hd + 1] = psum; }
}

// (round 6: the chunk weights w_s = exp(m_s - M) are computed ONCE, by thread s, into LDS - every one of the 128 threads evaluated all of them in
// f64 before, twice: 8.8 us of a 12.6 + 8.8 us attention at T = 1000, profiles/r06_longctx.txt.  Same values, same order of the sums.)
constexpr int ATT_COMBINE_MAX = 256;   // chunks a combine can weigh in LDS (ctx up to 32768)
// (after the evidence session: the loops over the chunks fetch EIGHT partials at a time before they use them - one thread walked its column chunk by chunk, a
// dependent L2 round trip each: 8.6 us of a 15.8 + 8.6 us attention at T = 2000.  Same values, same order of every sum.)
__global__ __launch_bounds__(128) void k_attention_combine(const AttnArgs a, const float* __restrict__ part, uint32_t nch) {
    LH_TOUCH_ARGS(a.q, a.sp, a.rows, nch);
    __shared__ float wsh[ATT_COMBINE_MAX];
    const uint32_t h = blockIdx.x, hd = a.hd, j = blockIdx.y;
    const uint32_t past = a.rows ? a.rows[j].pos : (a.sp ? a.sp->past : a.past_host);
    const uint32_t T = past + 1, n = (T + ATT_TC - 1) / ATT_TC;  // active chunks
    const float* base = part + ((size_t)j * gridDim.x + h) * nch * (hd + 2);
    const size_t st = hd + 2;
    constexpr uint32_t PF = 8;
    float M = -INFINITY;
    for (uint32_t s0 = 0; s0 < n; s0 += PF) {
        float v[PF];
#pragma unroll
        for (uint32_t u = 0; u < PF; ++u) v[u] = s0 + u < n ? base[(size_t)(s0 + u) * st + hd] : -INFINITY;
#pragma unroll
        for (uint32_t u = 0; u < PF; ++u) M = fmaxf(M, v[u]);
    }
    const bool lds_w = n <= (uint32_t)ATT_COMBINE_MAX;
    if (lds_w) {
        for (uint32_t s = threadIdx.x; s < n; s += 128) wsh[s] = (float)exp((double)__fsub_rn(base[(size_t)s * st + hd], M));
        __syncthreads();
    }
    auto w_of = [&](uint32_t s) { return lds_w ? wsh[s] : (float)exp((double)__fsub_rn(base[(size_t)s * st + hd], M)); };
    float l = 0.f;
    for (uint32_t s0 = 0; s0 < n; s0 += PF) {
        float v[PF];
#pragma unroll
        for (uint32_t u = 0; u < PF; ++u) v[u] = s0 + u < n ? base[(size_t)(s0 + u) * st + hd + 1] : 0.f;
#pragma unroll
        for (uint32_t u = 0; u < PF; ++u) if (s0 + u < n) l = fmaf(v[u], w_of(s0 + u), l);
    }
    const float inv = __fdiv_rn(1.0f, l);
    for (uint32_t c = threadIdx.x; c < hd; c += 128) {
        float o = 0.f;
        for (uint32_t s0 = 0; s0 < n; s0 += PF) {
            float v[PF];
#pragma unroll
            for (uint32_t u = 0; u < PF; ++u) v[u] = s0 + u < n ? base[(size_t)(s0 + u) * st + c] : 0.f;
#pragma unroll
            for (uint32_t u = 0; u < PF; ++u) if (s0 + u < n) o = fmaf(v[u], w_of(s0 + u), o);
        }
        const float ov = __fmul_rn(o, inv);
        a.out[(size_t)j * a.d + h * hd + c] = ov;
        if (a.out_s3) attn_store_split3(a, (size_t)j * a.d + h * hd + c, ov);
    }
}

// ---------------------------------------------------------------------------------------------------
// Small kernels
// ---------------------------------------------------------------------------------------------------
// Step parameters travel as KERNEL ARGUMENTS into device memory (stream-ordered, nothing on the host to keep alive: an
// async copy from a pinned slot could be overwritten by a caller that runs many steps ahead of the GPU).
__global__ void k_set_step(StepParams* sp, uint32_t token, uint32_t past, uint32_t step) {
    sp->token = token;
    sp->past = past;
    sp->step = step;
    sp->pad = 0;
}

// GetRows ml.go:1711-1750 — embedding lookup; token ids from the device step parameters (decode) or a device array.
// Host-provided ids are validated before launch; an id that arrives through device memory (argmax of the previous step,
// or received from the last pipeline rank) is clamped so that a corrupted value can never gather outside the table.
__global__ __launch_bounds__(256) void k_embed(const float* __restrict__ emb, const uint32_t* __restrict__ tokens, const StepParams* sp,
                                                float* __restrict__ x, uint32_t d, uint32_t vocab) {
    const uint32_t row = blockIdx.x;
    uint32_t tok = tokens ? tokens[row] : sp->token;
    tok = tok < vocab ? tok : vocab - 1;
    const f4* src = (const f4*)(emb + (size_t)tok * d);
    f4* dst = (f4*)(x + (size_t)row * d);
    for (uint32_t i = threadIdx.x; i < d / 4; i += 256) dst[i] = src[i];
}

// Greedy argmax over the logits (strict >, lowest index on ties: SURVEY §8c) + advance of the resident loop.
__global__ __launch_bounds__(1024) void k_argmax_advance(const float* __restrict__ logits, uint32_t n, StepParams* sp, uint32_t* __restrict__ out_tokens,
                                                         uint32_t* __restrict__ argmax_out, int advance) {
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float bv = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    auto take = [&](float v, uint32_t i) {  // ascending i inside a thread: the first maximum is kept
        if (v > bv || bi == 0xFFFFFFFFu) { bv = v; bi = i; }
    };
    const uint32_t n4 = (n % 4 == 0 && ((uintptr_t)logits & 15) == 0) ? n / 4 : 0;
    for (uint32_t i0 = tid; i0 < n4; i0 += 1024 * 4) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * 1024;
            v[u] = i < n4 ? ((const f4*)logits)[i] : f4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * 1024;
            if (i < n4) { take(v[u].x, 4 * i); take(v[u].y, 4 * i + 1); take(v[u].z, 4 * i + 2); take(v[u].w, 4 * i + 3); }
        }
    }
    for (uint32_t i = 4 * n4 + tid; i < n; i += 1024) take(logits[i], i);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const uint32_t oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        if (argmax_out) *argmax_out = bi;
        if (advance) {
            out_tokens[sp->step] = bi;
            sp->token = bi;
            sp->past += 1;
            sp->step += 1;
        }
    }
}

// RMSNorm + weight for N rows (prefill): one workgroup per row (ml.go:1753-1812 then 1877-1914: fp32 squares, f64 sum, one fp32 scale,
// two roundings per element).  Rows of up to 8192 floats (multiple of 4, 16-byte aligned) stay in registers between the two passes
// with all their loads in flight at once: the first version walked the row 256 floats at a time, one dependent L2 round trip per
// step, twice - 12.8 us per launch on 7B, 65 launches per prompt (profiles/r02b_ttft_kernel_trace.txt).
__global__ __launch_bounds__(256) void k_rmsnorm_rows(const float* __restrict__ x, const float* __restrict__ gamma, float* __restrict__ y, uint32_t d) {
    __shared__ double sred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (size_t)blockIdx.x * d;
    float* yr = y + (size_t)blockIdx.x * d;
    constexpr int NV = 8;
    const bool vec = d % 4 == 0 && d <= 256u * 4 * NV && (((uintptr_t)xr | (uintptr_t)yr | (uintptr_t)gamma) & 15) == 0;
    if (vec) {
        const uint32_t d4 = d / 4;
        f4 v[NV], g[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {   // unconditional loads from a clamped index (a branch around a load costs a full drain)
            const uint32_t i = (uint32_t)tid + (uint32_t)j * 256;
            v[j] = ((const f4*)xr)[i < d4 ? i : 0];
            g[j] = gamma ? ((const f4*)gamma)[i < d4 ? i : 0] : f4{1.f, 1.f, 1.f, 1.f};
        }
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if ((uint32_t)tid + (uint32_t)j * 256 < d4) {
                s += (double)__fmul_rn(v[j].x, v[j].x); s += (double)__fmul_rn(v[j].y, v[j].y);
                s += (double)__fmul_rn(v[j].z, v[j].z); s += (double)__fmul_rn(v[j].w, v[j].w);
            }
        s = wave_sum_f64(s);
        if (lane == 0) sred[wave] = s;
        __syncthreads();
        const double mean = (((sred[0] + sred[1]) + sred[2]) + sred[3]) / (double)d;
        const float scale = (float)(1.0 / sqrt(mean + 1e-5));
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const uint32_t i = (uint32_t)tid + (uint32_t)j * 256;
            if (i < d4) {
                f4 o;
                o.x = __fmul_rn(v[j].x, scale); o.y = __fmul_rn(v[j].y, scale); o.z = __fmul_rn(v[j].z, scale); o.w = __fmul_rn(v[j].w, scale);
                if (gamma) { o.x = __fmul_rn(g[j].x, o.x); o.y = __fmul_rn(g[j].y, o.y); o.z = __fmul_rn(g[j].z, o.z); o.w = __fmul_rn(g[j].w, o.w); }
                ((f4*)yr)[i] = o;
            }
        }
        return;
    }
    double s = 0.0;
    for (uint32_t i = tid; i < d; i += 256) s += (double)__fmul_rn(xr[i], xr[i]);
    s = wave_sum_f64(s);
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    const double mean = (((sred[0] + sred[1]) + sred[2]) + sred[3]) / (double)d;
    const float scale = (float)(1.0 / sqrt(mean + 1e-5));
    for (uint32_t i = tid; i < d; i += 256) {
        const float t = __fmul_rn(xr[i], scale);
        yr[i] = gamma ? __fmul_rn(gamma[i], t) : t;
    }
}

// RoPE on Q (mode 0) and on the new K rows, K/V append into the cache (prefill): qkv rows [N][3][d] -> q [N][d], caches.
__global__ __launch_bounds__(256) void k_rope_store(const float* __restrict__ qraw, const float* __restrict__ kraw, const float* __restrict__ vraw,
                                                     float* __restrict__ q, float* __restrict__ k_cache, float* __restrict__ v_cache,
                                                     const double2* __restrict__ rope, uint32_t d, uint32_t hd, uint32_t past,
                                                     const BatchRow* __restrict__ rows = nullptr, uint64_t kv_off = 0) {
    const uint32_t row = blockIdx.x, pos = rows ? rows[row].pos : past + row;
    if (rows) { k_cache = rows[row].kc + kv_off; v_cache = rows[row].vc + kv_off; }   // batched Eval: the row's own cache
    for (uint32_t e = threadIdx.x * 2; e < d; e += 512) {
        const double2 cs = rope[(size_t)pos * (hd >> 1) + ((e % hd) >> 1)];
        float o0, o1;
        rope_rotate(qraw[(size_t)row * d + e], qraw[(size_t)row * d + e + 1], cs, &o0, &o1);
        q[(size_t)row * d + e] = o0;
        q[(size_t)row * d + e + 1] = o1;
        rope_rotate(kraw[(size_t)row * d + e], kraw[(size_t)row * d + e + 1], cs, &o0, &o1);
        k_cache[(size_t)pos * d + e] = o0;
        k_cache[(size_t)pos * d + e + 1] = o1;
        v_cache[(size_t)pos * d + e] = vraw[(size_t)row * d + e];
        v_cache[(size_t)pos * d + e + 1] = vraw[(size_t)row * d + e + 1];
    }
}

// ---- batched decode (lh_batch): bookkeeping of the row table ---------------------------------------------------------------------
// The table rows[] and the per-row StepParams sp[] (what the N = 1 decode kernels read when a batch runs row by row) move in lockstep:
// every kernel that advances one advances the other, so a tick needs no separate synchronisation launch.
// One thread per row: position, output index and (first stage) token of the next tick, from kernel arguments (<= 64 rows).
struct BatchSetArgs { uint32_t pos[64]; uint32_t tok[64]; };
__global__ void k_batch_set(BatchRow* rows, uint32_t* tok, uint32_t n, int set_tok, const BatchSetArgs v, StepParams* sp, uint32_t step0) {
    const uint32_t i = threadIdx.x;
    if (i >= n) return;
    rows[i].pos = v.pos[i];
    rows[i].step = step0;
    if (set_tok) tok[i] = v.tok[i];
    sp[i].token = set_tok ? v.tok[i] : tok[i];
    sp[i].past = v.pos[i];
    sp[i].step = step0;
    sp[i].pad = 0;
}
// Greedy argmax of every row's logits (strict >, lowest index on ties: SURVEY §8c) -> ids_out[row]; `advance`: the id becomes the row's
// next token, is appended to the row's output list out[row * out_cap + step] and the row moves on by one position (the resident loop).
// grid = rows, 1024 threads; a row is touched by its own workgroup only.
__global__ __launch_bounds__(1024) void k_batch_argmax(const float* __restrict__ logits, uint32_t V, BatchRow* rows, uint32_t* tok, uint32_t* __restrict__ ids_out,
                                                       uint32_t* __restrict__ out, uint32_t out_cap, StepParams* sp, int advance) {
    __shared__ float sv[16];
    __shared__ uint32_t si[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t row = blockIdx.x;
    const float* lg = logits + (size_t)row * V;
    float bv = -INFINITY;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = tid; i < V; i += 1024) {   // ascending i inside a thread: the first maximum is kept
        const float v = lg[i];
        if (v > bv || bi == 0xFFFFFFFFu) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const uint32_t oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { sv[wave] = bv; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        if (ids_out) ids_out[row] = bi;
        if (advance) {
            const uint32_t st = rows[row].step;
            if (out && st < out_cap) out[(size_t)row * out_cap + st] = bi;
            tok[row] = bi;
            rows[row].pos += 1;
            rows[row].step = st + 1;
            sp[row].token = bi;
            sp[row].past += 1;
            sp[row].step = st + 1;
        }
    }
}
// every row moves on by one position (stages that do not produce ids: the last stage's argmax / sampler does it there)
__global__ void k_batch_advance(BatchRow* rows, StepParams* sp, uint32_t n) {
    const uint32_t i = threadIdx.x;
    if (i >= n) return;
    rows[i].pos += 1;
    rows[i].step += 1;
    sp[i].past += 1;
    sp[i].step += 1;
}
// lh_batch_set_sampler: the rows' output lists restart; ids and positions are untouched (rows[] and sp[] stay in lockstep)
__global__ void k_batch_reset_steps(BatchRow* rows, StepParams* sp, uint32_t n) {
    const uint32_t i = threadIdx.x;
    if (i < n) { rows[i].step = 0; sp[i].step = 0; }
}
// after the per-row sampler launches (each advanced its row's StepParams: token, past + 1, step + 1, and appended the id to the row's
// ring and output list): the row table follows
__global__ void k_batch_from_sp(BatchRow* rows, uint32_t* tok, uint32_t* ids, const StepParams* sp, uint32_t n) {
    const uint32_t i = threadIdx.x;
    if (i < n) { tok[i] = sp[i].token; ids[i] = sp[i].token; rows[i].pos += 1; rows[i].step += 1; }
}

// silu(a) * b elementwise (prefill FFN gate).
__global__ __launch_bounds__(256) void k_silu_mul(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    for (; i < n; i += stride) y[i] = __fmul_rn(silu_ref(a[i]), b[i]);
}

}  // namespace lh
