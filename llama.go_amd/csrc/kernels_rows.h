// csrc/kernels_rows.h — the decode weight stream for 2..4 activation rows at once (round 3).
//
// Why: the pods of a rank advance together in one pass over the weights (lh_batch; the reference runs them as independent goroutines,
// pkg/server/server.go:88-101), and prompts of 2..4 tokens are one Eval (server.go:185-192).  The 2..16-row launches of k_stream_mm2
// (kernels_stream.h) stream at 5.2-5.5 TB/s: their K-chunked LDS image makes a load instruction touch 512-byte segments of many rows.
// With two to four rows the arithmetic still fits the vector ALU next to the stream (4 x rows FMAs per 16-byte load: 13 of a CU's 64
// lanes per clock at four rows), so this kernel keeps k_gemv_sa's structure (kernels_llama.h) - one fat workgroup per CU, contiguous
// row blocks, whole weight rows streamed with scalar-addressed non-temporal loads, activations in registers, ONE barrier - and carries
// NC activation rows through it: NC accumulators per weight row, NC wave reductions, and the same fused prologue / epilogues per
// activation row (RMSNorm, RoPE + cache append at the row's own position of its own cache, silu * mul, + residual).
// Per activation row the arithmetic and the summation order are those of k_gemv_sa: a pod decodes bit-identical logits whether it
// runs alone (k_gemv_sa) or in a tick of two to four (this kernel).
#pragma once
#include "kernels_llama.h"
#include "kernels_q8.h"

namespace lh {

struct GemvRowsArgs {
    const float* w[3];      // matrix bases (MAP_BLOCK: [wq, wk, wv]; MAP_PAIR: [w1, w3]); block-int8: the int8 planes
    const float* ws[3];     // block-int8 only (k_gemv_q8_rows): per-32-column scales [rows][K / 32]
    uint32_t rows_per_mat;  // MAP_BLOCK
    uint32_t M, K;          // virtual weight rows, columns
    const float* x;         // activation rows [n][ldx]
    const float* gamma;     // PRO_RMSNORM: norm weight [K]
    float* y;               // EPI_STORE / EPI_RESID / EPI_SILU_MUL: [n][ldy]
    const float* resid;     // EPI_RESID: [n][ldy]
    float* q_out;           // EPI_QKV_ROPE: roped Q [n][d]
    float* k_cache;         //   prompt rows (rows == nullptr): this layer's slot [ctx][d], activation row r at position past + r
    float* v_cache;
    const double2* rope;    // [pos][hd / 2] (cos, sin)
    const BatchRow* rows;   //   rows of different streams: row r at position rows[r].pos of the cache rows[r].kc / .vc + kv_off
    uint64_t kv_off;
    uint32_t ldx, ldy, hd, d, past, n;   // n <= NC activation rows
    uint32_t wg_q, wg_r;    // (M / 2) / #workgroups and its remainder (wg_row_block; filled by launch_gemv_rows / launch_gemv_q8_rows)
};

template <int KI, int U, int TH, int NC, int PRO, int EPI, int MAP>
__global__ __launch_bounds__(TH) void k_gemv_rows(const GemvRowsArgs a) {
    LH_TOUCH_ARGS(a.w[0], a.x, a.rows, a.wg_q);   // the argument block's lines behind one wait (kernels_common.h)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NW = TH / 64;
    double* sred = (double*)smem_raw;                       // [NC][NW]
    float* red = (float*)(smem_raw + NC * NW * 8);          // [weight rows of this workgroup][NC][NW] per-wave partial dot products
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t K4 = a.K >> 2;
    uint32_t r0, r1;
    wg_row_block(a.M, a.wg_q, a.wg_r, &r0, &r1);
    const char *w0 = (const char*)a.w[0], *w1 = (const char*)a.w[1], *w2 = (const char*)a.w[2];
    const char* xdummy = (const char*)a.x;                  // K floats = exactly one row's extent: any lane offset stays inside it
    const uint32_t rpm = a.rows_per_mat;
    const uint64_t row_bytes = (uint64_t)a.K * 4;

    f4 xr[NC][KI];
    f4 gr[KI];
    bool act[KI];
    uint32_t loff[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        act[j] = (uint32_t)(tid + j * TH) < K4;
        loff[j] = act[j] ? (uint32_t)(tid + j * TH) * 16u : 0u;
        if (PRO == PRO_RMSNORM) gr[j] = act[j] ? ((const f4*)a.gamma)[tid + j * TH] : f4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float* xc = a.x + (size_t)((uint32_t)c < a.n ? c : 0) * a.ldx;   // rows past n: a duplicate of row 0, results never stored
#pragma unroll
        for (int j = 0; j < KI; ++j) xr[c][j] = act[j] ? ((const f4*)xc)[tid + j * TH] : f4{0.f, 0.f, 0.f, 0.f};
    }
    f4 w[U][KI];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const char* p = (r0 + u < r1) ? gemv_row_base<MAP>(w0, w1, w2, rpm, r0 + u, row_bytes) : xdummy;
#pragma unroll
        for (int j = 0; j < KI; ++j) w[u][j] = ld_nt((const f4*)(p + loff[j]));
    }
    if (PRO == PRO_RMSNORM) {
        // k_gemv_sa's norm (rmsnorm_prologue) for every activation row, the NC reductions side by side and ONE barrier for all of them (row by
        // row cost a barrier and a dependent f64 reduction chain each: 8 of them in front of every norm launch of an eight-row tick); per row
        // the same squares, the same f64 lane / wave / cross-wave order, the same two roundings per element -> the same bits
        double ss[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double t = 0.0;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                if (act[j]) {
                    t += (double)__fmul_rn(xr[c][j].x, xr[c][j].x);
                    t += (double)__fmul_rn(xr[c][j].y, xr[c][j].y);
                    t += (double)__fmul_rn(xr[c][j].z, xr[c][j].z);
                    t += (double)__fmul_rn(xr[c][j].w, xr[c][j].w);
                }
            }
            ss[c] = t;
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) ss[c] = wave_sum_f64(ss[c]);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) sred[c * NW + wave] = ss[c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double tot = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) tot += sred[c * NW + w2];
            const float scale = (float)(1.0 / sqrt(tot / (double)a.K + 1e-5));
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                if (act[j]) {
                    const f4 g = gr[j];
                    xr[c][j].x = __fmul_rn(g.x, __fmul_rn(xr[c][j].x, scale));
                    xr[c][j].y = __fmul_rn(g.y, __fmul_rn(xr[c][j].y, scale));
                    xr[c][j].z = __fmul_rn(g.z, __fmul_rn(xr[c][j].z, scale));
                    xr[c][j].w = __fmul_rn(g.w, __fmul_rn(xr[c][j].w, scale));
                }
            }
        }
    }
    for (uint32_t r = r0; r < r1; r += U) {
        float acc[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t nr = r + U + u;
            const char* p = nr < r1 ? gemv_row_base<MAP>(w0, w1, w2, rpm, nr, row_bytes) : xdummy;
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[u][c] = 0.f;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const f4 q = w[u][j];
#pragma unroll
                for (int c = 0; c < NC; ++c) {      // per activation row the chain of k_gemv_sa: x, y, z, w of one float4 after the other
                    float s = acc[u][c];
                    s = fmaf(q.x, xr[c][j].x, s);
                    s = fmaf(q.y, xr[c][j].y, s);
                    s = fmaf(q.z, xr[c][j].z, s);
                    s = fmaf(q.w, xr[c][j].w, s);
                    acc[u][c] = s;
                }
                w[u][j] = ld_nt((const f4*)(p + loff[j]));
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[u][c] = wave_sum(acc[u][c]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (r + u < r1) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) red[((size_t)(r - r0 + u) * NC + c) * NW + wave] = acc[u][c];
                }
        }
    }
    __syncthreads();
    // ---- epilogue: work item (weight row or pair `it`, activation row c) -> thread, all items side by side (a thread per weight row looping over
    // the activation rows left most of the workgroup idle behind up to eight silu / RoPE evaluations in a row); cross-wave sums in wave order
    // (bit-reproducible, the order of gemv_finish)
    constexpr uint32_t PER = (EPI == EPI_STORE || EPI == EPI_RESID) ? 1u : 2u;
    const uint32_t nit = (r1 - r0 + PER - 1) / PER, nwork = nit * a.n;
    for (uint32_t wk = (uint32_t)tid; wk < nwork; wk += TH) {
        const uint32_t c = wk / nit, fin = (wk - c * nit) * PER;
        const uint32_t v = r0 + fin;
        const float* p0 = red + ((size_t)fin * NC + c) * NW;
        float s0 = 0.f;
#pragma unroll
        for (int k = 0; k < NW; ++k) s0 += p0[k];
        if (EPI == EPI_STORE) {
            a.y[(size_t)c * a.ldy + v] = s0;
        } else if (EPI == EPI_RESID) {
            a.y[(size_t)c * a.ldy + v] = __fadd_rn(s0, a.resid[(size_t)c * a.ldy + v]);   // Add(cur, inp) ml.go:2515-2584
        } else {
            const float* p1 = p0 + (size_t)NC * NW;      // the pair's second weight row
            float s1 = 0.f;
#pragma unroll
            for (int k = 0; k < NW; ++k) s1 += p1[k];
            if (EPI == EPI_SILU_MUL) {
                a.y[(size_t)c * a.ldy + (v >> 1)] = __fmul_rn(silu_ref(s0), s1);   // ml.go:2587-2589, 1877-1914 (llama.go:354-361)
            } else {   // EPI_QKV_ROPE: Rope mode 0 on Q / mode 1 on the new K row (ml.go:2253-2328), K, V appended to the row's cache (llama.go:274-278)
                const uint32_t d = a.d;
                const uint32_t pos = a.rows ? a.rows[c].pos : a.past + c;
                float* kcb = a.rows ? a.rows[c].kc + a.kv_off : a.k_cache;
                float* vcb = a.rows ? a.rows[c].vc + a.kv_off : a.v_cache;
                if (v < 2 * d) {
                    const uint32_t e = v < d ? v : v - d;
                    const double2 cs = a.rope[(size_t)pos * (a.hd >> 1) + ((e % a.hd) >> 1)];
                    float o0, o1;
                    rope_rotate(s0, s1, cs, &o0, &o1);
                    float* dst = v < d ? a.q_out + (size_t)c * d + e : kcb + (size_t)pos * d + e;
                    dst[0] = o0;
                    dst[1] = o1;
                } else {
                    float* dst = vcb + (size_t)pos * d + (v - 2 * d);
                    dst[0] = s0;
                    dst[1] = s1;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same for block-int8 weights: k_gemv_q8s (kernels_q8.h) with NC activation rows.  A 16-byte load = 16 quants is converted ONCE
// (16 v_cvt) and multiplied into every row's pair of packed accumulators (8 v_pk_fma per row), the block scale is applied per row as in
// the single-row kernel; per activation row the arithmetic and its order are those of k_gemv_q8s (dot16_q8, then fmaf with the scale,
// DPP wave sum, cross-wave sum in wave order): a pod's int8 logits are bit-identical alone and in a tick of two to four.
// Launch shape = the single-row launch's (gemv_q8 in plan.hip): 256-thread workgroups, 256 threads per row, KI 16-quant chunks per thread
// (KI = 1: K <= 4096; KI = 3: 8192 < K <= 12288) - the same per-lane partition of a row, hence the same sums.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int KI, int U, int TPR, int TH, int NC, int PRO, int EPI, int MAP>
__global__ __launch_bounds__(TH) void k_gemv_q8_rows(const GemvRowsArgs a) {
    LH_TOUCH_ARGS(a.w[0], a.x, a.rows, a.wg_q);   // the argument block's lines behind one wait (kernels_common.h)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int G = TH / TPR, NWR = TPR / 64;
    static_assert(TPR % 64 == 0, "a row group must be a whole number of waves");
    double* sred = (double*)smem_raw;                        // [NC][16]
    float* red = (float*)(smem_raw + NC * 16 * 8);           // [weight rows of this workgroup][NC][NWR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tr = tid % TPR, wr = wave % NWR;
    const uint32_t grp = (uint32_t)__builtin_amdgcn_readfirstlane(tid / TPR);   // uniform within a wave
    const uint32_t K = a.K, K16 = K >> 4;
    uint32_t r0, r1;
    wg_row_block(a.M, a.wg_q, a.wg_r, &r0, &r1);
    // matrix bases as scalar integers + distances (see k_gemv_q8s)
    const uint64_t q0 = (uint64_t)sgpr_ptr(a.w[0]), s0 = (uint64_t)sgpr_ptr(a.ws[0]);
    const uint64_t dq1 = MAP == MAP_SINGLE ? 0 : (uint64_t)sgpr_ptr(a.w[1]) - q0, ds1 = MAP == MAP_SINGLE ? 0 : (uint64_t)sgpr_ptr(a.ws[1]) - s0;
    const uint64_t dq2 = MAP == MAP_BLOCK ? (uint64_t)sgpr_ptr(a.w[2]) - q0 - dq1 : 0, ds2 = MAP == MAP_BLOCK ? (uint64_t)sgpr_ptr(a.ws[2]) - s0 - ds1 : 0;
    const uint64_t xdummy = (uint64_t)sgpr_ptr(a.x);         // 4K bytes: covers a quant row (K bytes) and a scale row (K / 8 bytes)
    const uint32_t rpm = a.rows_per_mat;
    // MAP_BLOCK: virtual bases + the workgroup's matrix chosen once, one compare + select per row for the one boundary its rows may cross (see k_gemv_q8s)
    const uint64_t srow = (uint64_t)(K >> 5) * 4u;
    const uint64_t e1q = dq1 - (uint64_t)rpm * K, e2q = dq2 - (uint64_t)rpm * K, e1s = ds1 - (uint64_t)rpm * srow, e2s = ds2 - (uint64_t)rpm * srow;
    const uint32_t mlo = (r0 >= rpm ? 1u : 0u) + (r0 >= 2u * rpm ? 1u : 0u), mhi = (r1 - 1u >= rpm ? 1u : 0u) + (r1 - 1u >= 2u * rpm ? 1u : 0u);
    const bool one_mat = mhi - mlo <= 1u;
    const uint64_t qv_wg = q0 + (mlo >= 1u ? e1q : 0) + (mlo == 2u ? e2q : 0), sv_wg = s0 + (mlo >= 1u ? e1s : 0) + (mlo == 2u ? e2s : 0);
    const uint32_t bnd = (mlo + 1u) * rpm;
    const uint64_t nq = (mlo == 0u ? e1q : 0) + (mlo == 1u ? e2q : 0), ns = (mlo == 0u ? e1s : 0) + (mlo == 1u ? e2s : 0);
    bool act[KI];
    uint32_t qoff[KI], soff[KI];
    f4 xr[NC][KI][4];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        const uint32_t ch = (uint32_t)tr + (uint32_t)j * TPR;
        act[j] = ch < K16;
        qoff[j] = act[j] ? ch * 16u : 0u;
        soff[j] = act[j] ? (ch >> 1) * 4u : 0u;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float* xc = a.x + (size_t)((uint32_t)c < a.n ? c : 0) * a.ldx;
#pragma unroll
            for (int k = 0; k < 4; ++k) xr[c][j][k] = act[j] ? ((const f4*)xc)[ch * 4 + k] : f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    auto fetch = [&](u4 (&wd)[U][KI], float (&sd)[U][KI], uint32_t row_base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t row = row_base + G * u;
            uint64_t qb = xdummy, sb = xdummy;
            if (MAP == MAP_BLOCK) {
                if (row < r1) {
                    uint64_t bq = qv_wg + (row >= bnd ? nq : 0), bs = sv_wg + (row >= bnd ? ns : 0);
                    if (!one_mat) {
                        bq = q0 + (row >= rpm ? e1q : 0) + (row >= 2u * rpm ? e2q : 0);
                        bs = s0 + (row >= rpm ? e1s : 0) + (row >= 2u * rpm ? e2s : 0);
                    }
                    qb = bq + (uint64_t)row * K;
                    sb = bs + (uint64_t)row * srow;
                }
            } else
            if (row < r1) {   // scalar condition: s_cselect on the two addresses, no exec masking
                uint32_t m = 0, r = row;
                if (MAP == MAP_PAIR) { m = row & 1u; r = row >> 1; }
                qb = q0 + (m >= 1u ? dq1 : 0) + (uint64_t)r * K;
                sb = s0 + (m >= 1u ? ds1 : 0) + (uint64_t)r * srow;
            }
            typedef const u4 __attribute__((address_space(1))) gu4;
            typedef const float __attribute__((address_space(1))) gf32;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                wd[u][j] = __builtin_nontemporal_load((gu4*)(qb + qoff[j]));
                sd[u][j] = *(gf32*)(sb + soff[j]);
            }
        }
    };
    u4 wA[U][KI], wB[U][KI];
    float scA[U][KI], scB[U][KI];
    fetch(wA, scA, r0 + grp);

    if (PRO == PRO_RMSNORM) {
        f4 gr[KI][4];
#pragma unroll
        for (int j = 0; j < KI; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) gr[j][k] = act[j] ? ((const f4*)a.gamma)[((uint32_t)tr + (uint32_t)j * TPR) * 4 + k] : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < KI; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (act[j]) {
                        s += (double)__fmul_rn(xr[c][j][k].x, xr[c][j][k].x);
                        s += (double)__fmul_rn(xr[c][j][k].y, xr[c][j][k].y);
                        s += (double)__fmul_rn(xr[c][j][k].z, xr[c][j][k].z);
                        s += (double)__fmul_rn(xr[c][j][k].w, xr[c][j][k].w);
                    }
                }
            s = wave_sum_f64(s);
            if (lane == 0) sred[c * 16 + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double tot = 0.0;
#pragma unroll
            for (int k = 0; k < NWR; ++k) tot += sred[c * 16 + grp * NWR + k];
            const float scale = (float)(1.0 / sqrt(tot / (double)K + 1e-5));
#pragma unroll
            for (int j = 0; j < KI; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (act[j]) {
                        const f4 g = gr[j][k];
                        xr[c][j][k].x = __fmul_rn(g.x, __fmul_rn(xr[c][j][k].x, scale));
                        xr[c][j][k].y = __fmul_rn(g.y, __fmul_rn(xr[c][j][k].y, scale));
                        xr[c][j][k].z = __fmul_rn(g.z, __fmul_rn(xr[c][j][k].z, scale));
                        xr[c][j][k].w = __fmul_rn(g.w, __fmul_rn(xr[c][j][k].w, scale));
                    }
                }
        }
    }

    auto consume = [&](const u4 (&wd)[U][KI], const float (&sd)[U][KI], uint32_t rb) {
        float acc[U][NC];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[u][c] = 0.f;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                f2 w0[4], w1v[4];      // the 16 quants of the chunk as floats, converted once for all activation rows
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int dq = (int)wd[u][j][k];
                    w0[k] = f2{(float)(int)(signed char)(dq), (float)(int)(signed char)(dq >> 8)};
                    w1v[k] = f2{(float)(int)(signed char)(dq >> 16), (float)(dq >> 24)};
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {   // dot16_q8's two packed chains and their final adds, then the chunk's scale: k_gemv_q8s' order
                    f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        a0 = __builtin_elementwise_fma(w0[k], f2{xr[c][j][k].x, xr[c][j][k].y}, a0);
                        a1 = __builtin_elementwise_fma(w1v[k], f2{xr[c][j][k].z, xr[c][j][k].w}, a1);
                    }
                    acc[u][c] = fmaf(sd[u][j], (a0.x + a0.y) + (a1.x + a1.y), acc[u][c]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[u][c] = wave_sum_lane63(acc[u][c]);
        if (lane == 63) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t row = rb + G * u;
                if (row < r1) {
#pragma unroll
                    for (int c = 0; c < NC; ++c) red[((size_t)(row - r0) * NC + c) * NWR + wr] = acc[u][c];
                }
            }
        }
    };
    constexpr uint32_t STEP = G * U;
    for (uint32_t rb = r0 + grp; rb < r1; rb += 2 * STEP) {
        fetch(wB, scB, rb + STEP);
        consume(wA, scA, rb);
        fetch(wA, scA, rb + 2 * STEP);
        consume(wB, scB, rb + STEP);   // rows >= r1: loaded from the dummy, results dropped by the row test
    }
    __syncthreads();
    const uint32_t fin = (EPI == EPI_STORE || EPI == EPI_RESID) ? (uint32_t)tid : 2u * (uint32_t)tid;
    if (r0 + fin >= r1) return;
    const uint32_t v = r0 + fin;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if ((uint32_t)c >= a.n) break;
        const float* p0 = red + ((size_t)fin * NC + c) * NWR;
        float s0v = 0.f;
#pragma unroll
        for (int k = 0; k < NWR; ++k) s0v += p0[k];
        if (EPI == EPI_STORE) {
            a.y[(size_t)c * a.ldy + v] = s0v;
        } else if (EPI == EPI_RESID) {
            a.y[(size_t)c * a.ldy + v] = __fadd_rn(s0v, a.resid[(size_t)c * a.ldy + v]);
        } else {
            const float* p1 = p0 + (size_t)NC * NWR;
            float s1v = 0.f;
#pragma unroll
            for (int k = 0; k < NWR; ++k) s1v += p1[k];
            if (EPI == EPI_SILU_MUL) {
                a.y[(size_t)c * a.ldy + (v >> 1)] = __fmul_rn(silu_ref(s0v), s1v);
            } else {
                const uint32_t d = a.d;
                const uint32_t pos = a.rows ? a.rows[c].pos : a.past + (uint32_t)c;
                float* kcb = a.rows ? a.rows[c].kc + a.kv_off : a.k_cache;
                float* vcb = a.rows ? a.rows[c].vc + a.kv_off : a.v_cache;
                if (v < 2 * d) {
                    const uint32_t e = v < d ? v : v - d;
                    const double2 cs = a.rope[(size_t)pos * (a.hd >> 1) + ((e % a.hd) >> 1)];
                    float o0, o1;
                    rope_rotate(s0v, s1v, cs, &o0, &o1);
                    float* dst = v < d ? a.q_out + (size_t)c * d + e : kcb + (size_t)pos * d + e;
                    dst[0] = o0;
                    dst[1] = o1;
                } else {
                    float* dst = vcb + (size_t)pos * d + (v - 2 * d);
                    dst[0] = s0v;
                    dst[1] = s1v;
                }
            }
        }
    }
}

}  // namespace lh
