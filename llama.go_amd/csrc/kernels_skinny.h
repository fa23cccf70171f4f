// csrc/kernels_skinny.h — weight-streaming GEMM for SHORT prompts (2..8 token rows): one pass over the weights, fused like decode.
//
// Reference: server.Do feeds the whole prompt as ONE Eval (pkg/server/server.go:185-192), so time to first token for a short prompt
// is one pass over the weights: HBM-bound like decode (26.4 GB at 7B), not MFMA-bound.  Round 1 ran N <= 8 as five separate
// `gemm_small_n` launches per layer with norm / RoPE / SiLU as their own kernels, w2 streamed twice, and the 8-column register tile
// of k_gemv_cols spent more VALU time on its U*NC wave reductions than on FMAs (1.98 TB/s; 10.0 ms for 8 tokens vs a 3.3 ms floor).
//
// Here the contraction runs on the matrix cores purely as a REDUCTION ENGINE (MFMA only for dense W x contractions):
//   v_mfma_f32_16x16x4_f32   D[16 weight rows][16 token columns] += A[16 rows][4 k] * B[4 k][16 columns]
// accumulates over k inside the accumulator, so there is no cross-lane reduction, no LDS partial sum and no barrier in the stream.
// At HBM rate the matrix pipe is ~1/3 busy.
//   - grid = #CU workgroups of 4 waves; a workgroup owns a contiguous block of weight rows (the same split as k_gemv), cut into tiles
//     of 16 rows; a UNIT of work is (tile, K-range): with fewer tiles than waves (wo / w2 give a CU 16 rows = ONE tile) the
//     contraction is split 2 or 4 ways so that every wave streams — one wave alone is latency-bound at ~11 GB/s, a CU needs 25;
//     units are dealt to the waves round-robin, each wave parks its finished partial tiles in LDS, and after ONE barrier at the end
//     of the kernel thread (m, c) adds the partial sums of element (row m, column c) in fixed order and runs the epilogue;
//   - weights go global -> registers with non-temporal 16-byte loads, lane = (row l % 16, k-group l / 16) holds exactly the A
//     operands of 8 consecutive MFMAs; a ring of SK_RING k-blocks (2 KB each) per wave stays in flight ACROSS tile boundaries;
//   - the activation rows live in LDS (<= 8 x 4096 floats), staged once per launch (all loads of the stage in flight together) with
//     the RMSNorm*gamma prologue applied on the way (ml.go:1753-1812, 1877-1914); contractions longer than the LDS tile run as
//     several launches over K-chunks with raw partial sums handed through HBM (sequential launches, fixed order: deterministic);
//   - epilogue operands (residual rows, partial sums of earlier K-chunks, RoPE cos/sin) are staged in LDS up front;
//   - epilogues as in decode: residual add, SiLU*mul on (w1, w3) row pairs, RoPE + K/V cache append on (q, k, v) rows.
// Summation order differs from the scalar reference (k interleaved by groups of 4): within 1e-4 like every MFMA path.
#pragma once
#include "kernels_llama.h"

namespace lh {

typedef float f4v __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
    const float* w[3];      // matrix bases (MAP_BLOCK: [wq,wk,wv]; MAP_PAIR: [w1,w3])
    uint32_t rows_per_mat;  // MAP_BLOCK
    uint32_t M;             // virtual rows
    uint32_t K;             // full contraction length (row stride of the matrices, in floats)
    uint32_t k0, kc;        // this launch contracts columns [k0, k0 + kc), kc % 256 == 0
    const float* x;         // activations [n][K], row c at x + c * ldx
    uint32_t ldx, n;        // n <= NP token rows
    const float* gamma;     // PRO_RMSNORM
    const float* part_in;   // raw partial sums of the previous K-chunks [n][M], or NULL
    float* part_out;        // not the last chunk: store raw sums here instead of running the epilogue
    float* y;               // EPI_STORE / EPI_RESID: y[c * ldy + row]; EPI_SILU_MUL: y[c * ldy + row / 2]
    const float* resid;     // EPI_RESID, same layout as y
    uint32_t ldy;
    float* q_out;           // EPI_QKV_ROPE: roped Q [n][d]
    float* k_cache;         // this layer's K slot base [ctx][d]
    float* v_cache;
    const double2* rope;    // [pos][hd/2]
    uint32_t hd, d, past;
    uint32_t rows_cap;      // LDS capacity (rows) of the staged epilogue operands: >= rows of any workgroup
};

constexpr int SK_TH = 256, SK_NW = 4, SK_RING = 8, SK_KB = 32;   // a chunk is a whole number of ring groups: kc % (SK_RING * SK_KB) == 0

// LDS bytes of one launch (host and device agree through this one function)
// K-split of a workgroup's tiles: as many (tile, K-range) units as it takes to give all four waves the same amount of work
__host__ __device__ inline uint32_t skinny_ksplit(uint32_t ntiles) {
    return ntiles >= 8 || ntiles % 4 == 0 ? 1u : (ntiles % 2 == 0 ? 2u : 4u);
}
__host__ __device__ inline size_t skinny_lds_bytes(uint32_t NP, uint32_t kc, uint32_t rows_cap, uint32_t hd, bool resid, bool part, bool rope) {
    size_t b = (size_t)NP * (kc + 4) * 4;             // activation tile
    b += (size_t)NP * SK_NW * 8;                      // norm reduction
    const uint32_t nt = (rows_cap + 15) / 16;
    b += (size_t)(nt >= 8 ? nt : 28) * 16 * NP * 4;   // partial tiles of the units: 16 rows x NP columns each (<= 7 tiles x 4 ranges below 8 tiles)
    if (resid) b += (size_t)rows_cap * NP * 4;
    if (part) b += (size_t)rows_cap * NP * 4;
    if (rope) b += (size_t)NP * (hd / 2) * 16;
    return b + 64;
}

// Row base of virtual row v.  The matrix choice is arithmetic on the DISTANCES between the bases (selects against the constant 0):
// a select among three pointer values gets folded into an indexed read of a table in scratch memory (kernels_q8.h, round 2).
template <int MAP>
__device__ __forceinline__ const float* skinny_row(const SkinnyArgs& a, uint32_t v) {
    const uint64_t b0 = (uint64_t)a.w[0];
    if (MAP == MAP_SINGLE) return (const float*)b0 + (size_t)v * a.K;
    const uint64_t d1 = (uint64_t)a.w[1] - b0;
    if (MAP == MAP_BLOCK) {
        const uint64_t d2 = (uint64_t)a.w[2] - (uint64_t)a.w[1];
        const uint32_t m = (v >= a.rows_per_mat ? 1u : 0u) + (v >= 2u * a.rows_per_mat ? 1u : 0u);
        const uint64_t b = b0 + (m >= 1u ? d1 : 0) + (m == 2u ? d2 : 0);
        return (const float*)b + (size_t)(v - m * a.rows_per_mat) * a.K;
    }
    return (const float*)(b0 + ((v & 1u) ? d1 : 0)) + (size_t)(v >> 1) * a.K;
}

template <int NP, int PRO, int EPI, int MAP>
__global__ __launch_bounds__(SK_TH) void k_skinny(const SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: tile counts and stream positions stay in SGPRs
    const uint32_t nwg = gridDim.x, npairs = a.M >> 1;
    const uint32_t r0 = 2u * (uint32_t)(((uint64_t)blockIdx.x * npairs) / nwg);
    const uint32_t r1 = (blockIdx.x + 1 == nwg) ? a.M : 2u * (uint32_t)(((uint64_t)(blockIdx.x + 1) * npairs) / nwg);
    if (r1 <= r0) return;                                 // more workgroups than row pairs (tiny models): nothing to do, no barrier owed
    const uint32_t kc = a.kc, xstride = kc + 4;           // floats per staged activation row (+16 B: spreads the columns over the banks)
    const bool has_resid = EPI == EPI_RESID && !a.part_out, has_part = a.part_in != nullptr, has_rope = EPI == EPI_QKV_ROPE && !a.part_out;
    float* xs = (float*)smem_raw;                         // [NP][xstride]
    double* sred = (double*)(xs + (size_t)NP * xstride);  // [NP][SK_NW]
    const uint32_t nrows = r1 - r0, ntiles = (nrows + 15) / 16;
    const uint32_t ntcap = (a.rows_cap + 15) / 16;
    float* red = (float*)(sred + NP * SK_NW);             // [units][16 rows][NP] partial tiles
    float* res_s = red + (size_t)(ntcap >= 8 ? ntcap : 28) * 16 * NP;   // [rows][NP] residual of this workgroup's rows
    float* part_s = res_s + (has_resid ? (size_t)a.rows_cap * NP : 0);
    double2* rope_s = (double2*)(part_s + (has_part ? (size_t)a.rows_cap * NP : 0));   // [NP][hd/2]
    const uint32_t ngr = kc / (SK_RING * SK_KB);          // ring groups (256 floats of every row) in this launch's chunk
    const uint32_t S = skinny_ksplit(ntiles) <= ngr ? skinny_ksplit(ntiles) : 1u;   // K ranges per tile
    const uint32_t nunits = ntiles * S;                   // unit u = (tile u / S, range u % S), range r = groups [r ngr / S, (r+1) ngr / S)
    const uint32_t myunits = nunits > (uint32_t)wave ? (nunits - wave + SK_NW - 1) / SK_NW : 0;   // units wave, wave + 4, ...
    const uint32_t lm = lane & 15, lg = lane >> 4;        // lane = (row within tile / token column, k-group)

    // ---- weight stream.  Everything inside a group is straight-line code: a branch between a load and its use makes the compiler give up
    // counting (s_waitcnt vmcnt(0)) and the ring collapses to one load at a time (seen in the ISA).
    typedef const f4 __attribute__((address_space(1))) gf4;   // addresses are rebuilt from integers: say GLOBAL, or the loads become flat_load
    auto unit_g0 = [&](uint32_t u) -> uint32_t { return (u % S) * ngr / S; };
    auto unit_g1 = [&](uint32_t u) -> uint32_t { return (u % S + 1) * ngr / S; };
    auto unit_ptr = [&](uint32_t u) -> const float* {     // this lane's row of unit u, at the start of the unit's K range
        uint32_t row = r0 + (u / S) * 16 + lm;
        row = row < r1 ? row : r1 - 1;                    // partial last tile: duplicates of the last row, dropped in the epilogue
        return skinny_row<MAP>(a, row) + a.k0 + (size_t)unit_g0(u) * (SK_RING * SK_KB) + lg * 8;
    };
    // Two register sets of one group each: while a group is multiplied out of one set, the other set's group is already in flight and
    // the first set is refilled with the group after that (32 KB per wave requested ahead).  The load stream runs over this wave's
    // (unit, group) sequence two groups ahead of the multiplication.
    uint32_t total_groups = 0;
    for (uint32_t i = 0; i < myunits; ++i) total_groups += unit_g1(wave + SK_NW * i) - unit_g0(wave + SK_NW * i);
    uint32_t lunit = 0, lgrp = 0, lcount = myunits ? unit_g1(wave) - unit_g0(wave) : 1;   // load stream: unit index (of mine), group in it, groups in it
    const float* ltp = unit_ptr(myunits ? (uint32_t)wave : 0u);   // (waves without a unit load a valid row and never consume it)
    auto next_group_ptr = [&]() -> const float* {          // pointer of the load stream's next group, then advance (past the end: stays on the last group)
        const float* p = ltp + (size_t)lgrp * (SK_RING * SK_KB);
        if (lunit < myunits) {
            if (++lgrp == lcount) {
                ++lunit;
                if (lunit < myunits) {
                    const uint32_t u = wave + SK_NW * lunit;
                    lgrp = 0; lcount = unit_g1(u) - unit_g0(u); ltp = unit_ptr(u);
                } else {
                    lgrp = lcount - 1;
                }
            }
        }
        return p;
    };
    // ---- stage the activation rows (whole workgroup): all loads of a pass in flight together, RMSNorm * gamma on the way
    constexpr int NB = 4;                                 // float4 per thread per column per pass = 4096 floats per column per pass
    float scale[NP];
#pragma unroll
    for (int c = 0; c < NP; ++c) scale[c] = 1.f;
    const bool one_pass = a.K == kc && a.K <= (uint32_t)(SK_TH * 4 * NB);   // the staged chunk IS the whole row: loaded once
    f4 keep[NP][NB];
    if (PRO == PRO_RMSNORM) {                             // statistics over the FULL row, whatever this launch's chunk is
        double ss[NP];
#pragma unroll
        for (int c = 0; c < NP; ++c) ss[c] = 0.0;
        for (uint32_t kb0 = 0; kb0 < a.K; kb0 += SK_TH * 4 * NB) {
#pragma unroll
            for (int c = 0; c < NP; ++c)
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const uint32_t k = kb0 + (uint32_t)(i * SK_TH + tid) * 4;   // unconditional load from a clamped address, then select
                    const f4 v = *(const f4*)(a.x + (size_t)((uint32_t)c < a.n ? c : 0) * a.ldx + (k < a.K ? k : 0));
                    keep[c][i] = ((uint32_t)c < a.n && k < a.K) ? v : f4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
            for (int c = 0; c < NP; ++c)
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const f4 v = keep[c][i];
                    ss[c] += (double)__fmul_rn(v.x, v.x); ss[c] += (double)__fmul_rn(v.y, v.y); ss[c] += (double)__fmul_rn(v.z, v.z); ss[c] += (double)__fmul_rn(v.w, v.w);
                }
        }
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            const double s = wave_sum_f64(ss[c]);
            if (lane == 0) sred[c * SK_NW + wave] = s;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NP; ++c) {
            const double mean = (((sred[c * SK_NW] + sred[c * SK_NW + 1]) + sred[c * SK_NW + 2]) + sred[c * SK_NW + 3]) / (double)a.K;
            scale[c] = (float)(1.0 / sqrt(mean + 1e-5));
        }
    }
    for (uint32_t kb0 = 0; kb0 < kc; kb0 += SK_TH * 4 * NB) {
        if (!(PRO == PRO_RMSNORM && one_pass)) {
#pragma unroll
            for (int c = 0; c < NP; ++c)
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const uint32_t k = kb0 + (uint32_t)(i * SK_TH + tid) * 4;
                    const f4 v = *(const f4*)(a.x + (size_t)((uint32_t)c < a.n ? c : 0) * a.ldx + a.k0 + (k < kc ? k : 0));
                    keep[c][i] = ((uint32_t)c < a.n && k < kc) ? v : f4{0.f, 0.f, 0.f, 0.f};
                }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const uint32_t k = kb0 + (uint32_t)(i * SK_TH + tid) * 4;
            if (k < kc) {
                f4 g = {1.f, 1.f, 1.f, 1.f};
                if (PRO == PRO_RMSNORM) g = *(const f4*)(a.gamma + a.k0 + k);
#pragma unroll
                for (int c = 0; c < NP; ++c) {
                    f4 v = keep[c][i];
                    if (PRO == PRO_RMSNORM) {
                        v.x = __fmul_rn(g.x, __fmul_rn(v.x, scale[c])); v.y = __fmul_rn(g.y, __fmul_rn(v.y, scale[c]));
                        v.z = __fmul_rn(g.z, __fmul_rn(v.z, scale[c])); v.w = __fmul_rn(g.w, __fmul_rn(v.w, scale[c]));
                    }
                    *(f4*)(xs + (size_t)c * xstride + k) = v;   // columns >= n hold zeros (their results are never read)
                }
            }
        }
    }
    // ---- stage the epilogue operands of this workgroup's rows
    if (has_resid || has_part) {
        for (uint32_t e = tid; e < nrows * NP; e += SK_TH) {
            const uint32_t c = e / nrows, rr = e - c * nrows;   // consecutive threads -> consecutive rows of one column (coalesced)
            if (c < a.n) {
                if (has_resid) res_s[rr * NP + c] = a.resid[(size_t)c * a.ldy + r0 + rr];
                if (has_part) part_s[rr * NP + c] = a.part_in[(size_t)c * a.M + r0 + rr];
            }
        }
    }
    if (has_rope) {
        const uint32_t half = a.hd >> 1;
        for (uint32_t e = tid; e < a.n * half; e += SK_TH) {
            const uint32_t c = e / half, i = e - c * half;
            rope_s[c * half + i] = a.rope[(size_t)(a.past + c) * half + i];
        }
    }
    // The first two groups are requested only now: the vector-memory counter is in-order, so weight loads issued BEFORE the staging loads
    // above would have been drained by the first wait on a staged value anyway — and with them still pending at loop entry the compiler
    // protected the loop's LDS reads with near-draining waits on every iteration (seen in the ISA).
    f4 wa[SK_RING][2], wb[SK_RING][2];
    {
        const float* pa = next_group_ptr();
        const float* pb = next_group_ptr();
#pragma unroll
        for (int s = 0; s < SK_RING; ++s) {
            gf4* p = (gf4*)(uintptr_t)(pa + s * SK_KB);
            wa[s][0] = __builtin_nontemporal_load(p);
            wa[s][1] = __builtin_nontemporal_load(p + 1);
        }
#pragma unroll
        for (int s = 0; s < SK_RING; ++s) {
            gf4* p = (gf4*)(uintptr_t)(pb + s * SK_KB);
            wb[s][0] = __builtin_nontemporal_load(p);
            wb[s][1] = __builtin_nontemporal_load(p + 1);
        }
    }

    __syncthreads();

    // ---- main stream: no barrier, no cross-wave traffic
    const float* xl = xs + (size_t)(lm % NP) * xstride + lg * 8;   // this lane's B operands: column lm (mod NP), k-group lg
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    uint32_t cunit = 0, cgrp = 0;                          // multiplication position: unit index (of mine), group in it
    uint32_t cg0 = myunits ? unit_g0(wave) : 0, ccount = myunits ? unit_g1(wave) - cg0 : 1;
    // one group out of register set `w`, each slot refilled from `nb` (the group two ahead) right behind its last use
    auto run_group = [&](f4 (&w)[SK_RING][2], const float* nb) {
        const float* xg = xl + (size_t)(cg0 + cgrp) * (SK_RING * SK_KB);
#pragma unroll
        for (int s = 0; s < SK_RING; ++s) {
            const f4 w0 = w[s][0], w1 = w[s][1];
            const f4 x0 = *(const f4*)(xg + s * SK_KB), x1 = *(const f4*)(xg + s * SK_KB + 4);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, x0.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, x0.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, x0.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, x0.w, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, x1.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, x1.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, x1.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, x1.w, acc, 0, 0, 0);
            // the fences keep the scheduler from collecting all refills at the end of the group (it did: the set then drains with
            // vmcnt(0) every 16 KB)
            __builtin_amdgcn_sched_barrier(0);
            gf4* p = (gf4*)(uintptr_t)(nb + s * SK_KB);
            w[s][0] = __builtin_nontemporal_load(p);
            w[s][1] = __builtin_nontemporal_load(p + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto group_done = [&]() {
        if (++cgrp == ccount) {   // unit finished: lane (lg, lm) holds D[row 4 lg + i][column lm] of its K range -> LDS, columns < NP only
            const uint32_t u = wave + SK_NW * cunit;
            if (lm < (uint32_t)NP) {
                float* dst = red + (size_t)u * (16 * NP) + (lg * 4) * NP + lm;
                dst[0] = acc[0]; dst[NP] = acc[1]; dst[2 * NP] = acc[2]; dst[3 * NP] = acc[3];
            }
            acc = f4v{0.f, 0.f, 0.f, 0.f};
            cgrp = 0;
            ++cunit;
            if (cunit < myunits) { const uint32_t un = wave + SK_NW * cunit; cg0 = unit_g0(un); ccount = unit_g1(un) - cg0; }
        }
    };
    for (uint32_t G = 0; G < total_groups; G += 2) {
        run_group(wa, next_group_ptr());
        group_done();
        if (G + 1 < total_groups) {
            run_group(wb, next_group_ptr());
            group_done();
        }
    }
    __syncthreads();

    // ---- epilogue: thread (m, c) owns element (row m, column c) of every tile; K-range partial sums added in range order
    const uint32_t em = tid >> 4, ec = tid & 15;
    const bool pair_epi = (EPI == EPI_SILU_MUL || EPI == EPI_QKV_ROPE) && !a.part_out;
    if (ec < a.n && !(pair_epi && (em & 1))) {
        for (uint32_t t = 0; t < ntiles; ++t) {
            const uint32_t rl = t * 16 + em;              // row relative to r0
            if (rl >= nrows) break;
            auto elem = [&](uint32_t rr) {
                const float* e = red + (size_t)(t * S) * (16 * NP) + (rr - t * 16) * NP + ec;
                float v = e[0];
                for (uint32_t k = 1; k < S; ++k) v += e[(size_t)k * (16 * NP)];
                if (has_part) v += part_s[rr * NP + ec];
                return v;
            };
            const uint32_t row = r0 + rl, c = ec;
            const float s0 = elem(rl);
            if (a.part_out) {
                a.part_out[(size_t)c * a.M + row] = s0;
            } else if (EPI == EPI_STORE) {
                a.y[(size_t)c * a.ldy + row] = s0;
            } else if (EPI == EPI_RESID) {
                a.y[(size_t)c * a.ldy + row] = __fadd_rn(s0, res_s[rl * NP + c]);   // Add ml.go:2515-2584
            } else {
                const float s1 = elem(rl + 1);            // M is even and tiles start on even rows: the partner exists
                if (EPI == EPI_SILU_MUL) {                // Silu(w1 h) * (w3 h): ml.go:2587-2589, 1877-1914 (llama.go:354-361)
                    a.y[(size_t)c * a.ldy + (row >> 1)] = __fmul_rn(silu_ref(s0), s1);
                } else {                                  // RoPE on Q and the new K rows, K/V appended (llama.go:274-297)
                    const uint32_t d = a.d, pos = a.past + c, half = a.hd >> 1;
                    if (row < 2 * d) {
                        const uint32_t e = row < d ? row : row - d;
                        const double2 cs = rope_s[c * half + ((e % a.hd) >> 1)];
                        float o0, o1;
                        rope_rotate(s0, s1, cs, &o0, &o1);
                        float* dst = row < d ? a.q_out + (size_t)c * d + e : a.k_cache + (size_t)pos * d + e;
                        dst[0] = o0;
                        dst[1] = o1;
                    } else {
                        float* dst = a.v_cache + (size_t)pos * d + (row - 2 * d);
                        dst[0] = s0;
                        dst[1] = s1;
                    }
                }
            }
        }
    }
}

}  // namespace lh
