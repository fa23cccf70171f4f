// csrc/kernels_skinny.h — weight-streaming GEMM for SHORT prompts (2..8 token rows): one pass over the weights, fused like decode.
//
// Reference: server.Do feeds the whole prompt as ONE Eval (pkg/server/server.go:185-192), so time to first token for a short prompt
// is one pass over the weights: HBM-bound like decode (26.4 GB at 7B), not MFMA-bound.  Round 1 ran N <= 8 as five separate
// `gemm_small_n` launches per layer with norm / RoPE / SiLU as their own kernels, w2 streamed twice, and the 8-column register tile
// of k_gemv_cols spent more VALU time on its U*NC wave reductions than on FMAs (1.98 TB/s; 10.0 ms for 8 tokens vs a 3.3 ms floor).
//
// Here the contraction runs on the matrix cores purely as a REDUCTION ENGINE (MFMA only for dense W x contractions):
//   v_mfma_f32_4x4x1_16B_f32   16 independent blocks,  D_b[4 weight rows][4 token columns] += A_b[4 rows] (x) B_b[4 columns]
// block b takes the k's with k / 4 % 16 == b, every instruction advances all 16 blocks by one k, and the accumulators sum over k: per
// tile there is ONE cross-lane reduction (over the 16 blocks) at its end, no LDS partial sum and no barrier in the stream.
// Why this shape and not 16x16x4 (tried first, round 2): an A operand register holds lane <-> (row, k-slot), so one 16-byte load
// instruction covers R rows x (64 / R) lanes x 16 B.  tools/stream_probe measures what HBM makes of that on the 7B shapes: R = 16
// (64 contiguous bytes per row and instruction) 2.3-4.5 TB/s, R = 4 (256 B) 5.1-5.5 TB/s, R = 1 (the decode GEMV, 1 KB) 6.0-6.5 TB/s
// (profiles/r02_stream_pattern_probe.txt).  4x4x1 is the matrix instruction with the fewest rows per operand.
//   - grid = #CU workgroups of 4 waves; a workgroup owns a contiguous block of weight rows (the same split as k_gemv), cut into tiles
//     of 4 rows; a UNIT of work is (tile, K-range): when the tile count does not divide over the 4 waves the contraction is split 2 or
//     4 ways; units are dealt to the waves round-robin, each wave parks its finished partial tiles in LDS, and after ONE barrier at
//     the end of the kernel the threads add the partial sums of each element in fixed order and run the epilogue;
//   - weights go global -> registers with non-temporal 16-byte loads, lane = (row l % 16, k-group g = l / 16): per 32-float k-block
//     one load of floats 4g..4g+3 and one of 16+4g..16+4g+3, so that every load instruction covers whole 64-byte halves of the rows'
//     128-byte lines (k-groups 8 floats apart touched every line twice, half used each time); the two loads are exactly the A
//     operands of 8 consecutive MFMAs (any pairing of k's works as long as the B operand reads the activations the same way); a ring of SK_RING k-blocks (2 KB each) per wave stays in flight ACROSS tile boundaries;
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

constexpr int SK_TH = 256, SK_NW = 4, SK_RING = 4, SK_KI = 64, SK_TR = 4;   // a group = SK_RING instructions of SK_KI floats per row; a chunk is a whole number of groups
constexpr int SK_GRP = SK_RING * SK_KI;                                        // 256 floats of every row per group

// K-split of a workgroup's tiles: as many (tile, K-range) units as it takes to give all four waves the same amount of work
__host__ __device__ inline uint32_t skinny_ksplit(uint32_t ntiles) {
    return ntiles >= 16 || ntiles % 4 == 0 ? 1u : (ntiles % 2 == 0 ? 2u : 4u);
}
__host__ __device__ inline uint32_t skinny_units_cap(uint32_t rows_cap) {
    const uint32_t nt = (rows_cap + SK_TR - 1) / SK_TR;
    return nt > 60 ? nt : 60;                          // a workgroup with fewer rows than the cap may split: <= 15 tiles x 4 ranges below 16 tiles
}
// LDS bytes of one launch (host and device agree through this one function)
__host__ __device__ inline size_t skinny_lds_bytes(uint32_t NP, uint32_t kc, uint32_t rows_cap, uint32_t hd, bool resid, bool part, bool rope) {
    size_t b = (size_t)NP * (kc + 16) * 4;            // activation tile
    b += (size_t)NP * SK_NW * 8;                      // norm reduction
    b += (size_t)skinny_units_cap(rows_cap) * SK_TR * NP * 4;   // partial tiles of the units: 4 rows x NP columns each
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
    static_assert(NP == 8, "two column groups of four");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform on purpose: unit counts and stream positions stay in SGPRs
    const uint32_t nwg = gridDim.x, npairs = a.M >> 1;
    const uint32_t r0 = 2u * (uint32_t)(((uint64_t)blockIdx.x * npairs) / nwg);
    const uint32_t r1 = (blockIdx.x + 1 == nwg) ? a.M : 2u * (uint32_t)(((uint64_t)(blockIdx.x + 1) * npairs) / nwg);
    if (r1 <= r0) return;                                 // more workgroups than row pairs (tiny models): nothing to do, no barrier owed
    const uint32_t kc = a.kc, xstride = kc + 16;          // floats per staged activation row; +64 B: (column, block) pairs of 16 lanes hit 16 different 16-byte slots
    const bool has_resid = EPI == EPI_RESID && !a.part_out, has_part = a.part_in != nullptr, has_rope = EPI == EPI_QKV_ROPE && !a.part_out;
    float* xs = (float*)smem_raw;                         // [NP][xstride]
    double* sred = (double*)(xs + (size_t)NP * xstride);  // [NP][SK_NW]
    float* red = (float*)(sred + NP * SK_NW);             // [units][4 rows][NP] partial tiles
    float* res_s = red + (size_t)skinny_units_cap(a.rows_cap) * SK_TR * NP;   // [rows][NP] residual of this workgroup's rows
    float* part_s = res_s + (has_resid ? (size_t)a.rows_cap * NP : 0);
    double2* rope_s = (double2*)(part_s + (has_part ? (size_t)a.rows_cap * NP : 0));   // [NP][hd/2]
    const uint32_t nrows = r1 - r0, ntiles = (nrows + SK_TR - 1) / SK_TR;
    const uint32_t ngr = kc / SK_GRP;                     // groups in this launch's chunk
    const uint32_t S = skinny_ksplit(ntiles) <= ngr ? skinny_ksplit(ntiles) : 1u;   // K ranges per tile
    const uint32_t nunits = ntiles * S;                   // unit u = (tile u / S, range u % S), range r = groups [r ngr / S, (r+1) ngr / S)
    const uint32_t myunits = nunits > (uint32_t)wave ? (nunits - wave + SK_NW - 1) / SK_NW : 0;   // units wave, wave + 4, ...
    const uint32_t li = lane & 3, lb = lane >> 2;         // lane = (row within tile / token column within its group, block)

    // ---- weight stream.  Everything inside a group is straight-line code: a branch between a load and its use makes the compiler give up
    // counting (s_waitcnt vmcnt(0)) and the stream collapses to one load at a time (seen in the ISA).
    typedef const f4 __attribute__((address_space(1))) gf4;   // addresses are rebuilt from integers: say GLOBAL, or the loads become flat_load
    auto unit_g0 = [&](uint32_t u) -> uint32_t { return (u % S) * ngr / S; };
    auto unit_g1 = [&](uint32_t u) -> uint32_t { return (u % S + 1) * ngr / S; };
    auto unit_ptr = [&](uint32_t u) -> const float* {     // this lane's row of unit u, at the start of the unit's K range
        uint32_t row = r0 + (u / S) * SK_TR + li;
        row = row < r1 ? row : r1 - 1;                    // partial last tile: duplicates of the last row, dropped in the epilogue
        return skinny_row<MAP>(a, row) + a.k0 + (size_t)unit_g0(u) * SK_GRP + lb * 4;
    };
    uint32_t total_groups = 0;
    for (uint32_t i = 0; i < myunits; ++i) total_groups += unit_g1(wave + SK_NW * i) - unit_g0(wave + SK_NW * i);
    uint32_t lunit = 0, lgrp = 0, lcount = myunits ? unit_g1(wave) - unit_g0(wave) : 1;   // load stream: unit index (of mine), group in it, groups in it
    const float* ltp = myunits ? unit_ptr((uint32_t)wave) : a.x + lb * 4;   // (waves without a unit read the activations and never consume them)
    auto next_group_ptr = [&]() -> const float* {          // pointer of the load stream's next group, then advance
        const float* p = ltp + (size_t)lgrp * SK_GRP;
        if (lunit < myunits) {
            if (++lgrp == lcount) {
                ++lunit;
                if (lunit < myunits) {
                    const uint32_t u = wave + SK_NW * lunit;
                    lgrp = 0; lcount = unit_g1(u) - unit_g0(u); ltp = unit_ptr(u);
                } else {
                    lgrp = 0; lcount = 1; ltp = a.x + lb * 4;   // past the end: the (cache-resident) activations, never consumed
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
    // The first three groups are requested only now: the vector-memory counter is in-order, so weight loads issued BEFORE the staging
    // loads above would have been drained by the first wait on a staged value anyway — and with them still pending at loop entry the
    // compiler protected the loop's LDS reads with near-draining waits on every iteration (seen in the ISA).
    // FOUR register sets of one group each, every group requested as one BURST: the group being multiplied and three in flight
    // (12 KB per wave, 48 KB per CU ahead of the matrix pipe).
    f4 wa[SK_RING], wb[SK_RING], wc[SK_RING], wd[SK_RING];
    auto burst = [&](f4 (&w)[SK_RING], const float* nb) {
#pragma unroll
        for (int s = 0; s < SK_RING; ++s) w[s] = __builtin_nontemporal_load((gf4*)(uintptr_t)(nb + s * SK_KI));
    };
    burst(wa, next_group_ptr());
    burst(wb, next_group_ptr());
    burst(wc, next_group_ptr());
    __syncthreads();

    // ---- main stream: no barrier, no cross-wave traffic
    const float* xl = xs + (size_t)li * xstride + lb * 4;   // this lane's B operands: columns li and li + 4, block lb
    f4v acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // D_b[row v][column li] and D_b[row v][column li + 4]
    uint32_t cunit = 0, cgrp = 0;                          // multiplication position: unit index (of mine), group in it
    uint32_t cg0 = myunits ? unit_g0(wave) : 0, ccount = myunits ? unit_g1(wave) - cg0 : 1;
    auto run_group = [&](const f4 (&w)[SK_RING]) {         // one group out of register set `w` (straight-line)
        const float* xg = xl + (size_t)(cg0 + cgrp) * SK_GRP;
#pragma unroll
        for (int s = 0; s < SK_RING; ++s) {
            const f4 wv = w[s];
            const f4 x0 = *(const f4*)(xg + s * SK_KI), x1 = *(const f4*)(xg + s * SK_KI + 4 * xstride);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, x0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.x, x1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, x0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.y, x1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.z, x0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.z, x1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.w, x0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wv.w, x1.w, acc1, 0, 0, 0);
        }
    };
    auto group_done = [&]() {
        if (++cgrp == ccount) {   // unit finished: add the 16 blocks (lanes that differ in lb), lanes 0..3 park the 4 x 8 partial tile in LDS
            float v[8] = {acc0[0], acc0[1], acc0[2], acc0[3], acc1[0], acc1[1], acc1[2], acc1[3]};
#pragma unroll
            for (int o = 4; o < 64; o <<= 1)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] += __shfl_xor(v[k], o, 64);
            const uint32_t u = wave + SK_NW * cunit;
            if (lb == 0) {
                float* dst = red + (size_t)u * (SK_TR * NP) + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) { dst[r * NP] = v[r]; dst[r * NP + 4] = v[4 + r]; }
            }
            acc0 = f4v{0.f, 0.f, 0.f, 0.f};
            acc1 = f4v{0.f, 0.f, 0.f, 0.f};
            cgrp = 0;
            ++cunit;
            if (cunit < myunits) { const uint32_t un = wave + SK_NW * cunit; cg0 = unit_g0(un); ccount = unit_g1(un) - cg0; }
        }
    };
    for (uint32_t G = 0; G < total_groups; G += 4) {
        burst(wd, next_group_ptr());
        __builtin_amdgcn_sched_barrier(0);   // the burst stays a burst, ahead of the group it overlaps with
        run_group(wa);
        group_done();
        if (G + 1 < total_groups) {
            burst(wa, next_group_ptr());
            __builtin_amdgcn_sched_barrier(0);
            run_group(wb);
            group_done();
        }
        if (G + 2 < total_groups) {
            burst(wb, next_group_ptr());
            __builtin_amdgcn_sched_barrier(0);
            run_group(wc);
            group_done();
        }
        if (G + 3 < total_groups) {
            burst(wc, next_group_ptr());
            __builtin_amdgcn_sched_barrier(0);
            run_group(wd);
            group_done();
        }
    }
    __syncthreads();

    // ---- epilogue: one thread per element (or per row pair and column); K-range partial sums added in range order
    const bool pair_epi = (EPI == EPI_SILU_MUL || EPI == EPI_QKV_ROPE) && !a.part_out;
    const uint32_t nelem = (pair_epi ? nrows / 2 : nrows) * NP;
    for (uint32_t e = tid; e < nelem; e += SK_TH) {
        const uint32_t c = e % NP, rr = e / NP;
        if (c >= a.n) continue;
        const uint32_t rl = pair_epi ? 2 * rr : rr;       // row relative to r0
        auto elem = [&](uint32_t q) {
            const float* p = red + (size_t)((q / SK_TR) * S) * (SK_TR * NP) + (q % SK_TR) * NP + c;
            float v = p[0];
            for (uint32_t k = 1; k < S; ++k) v += p[(size_t)k * (SK_TR * NP)];
            if (has_part) v += part_s[q * NP + c];
            return v;
        };
        const uint32_t row = r0 + rl;
        const float s0 = elem(rl);
        if (a.part_out) {
            a.part_out[(size_t)c * a.M + row] = s0;
        } else if (EPI == EPI_STORE) {
            a.y[(size_t)c * a.ldy + row] = s0;
        } else if (EPI == EPI_RESID) {
            a.y[(size_t)c * a.ldy + row] = __fadd_rn(s0, res_s[rl * NP + c]);   // Add ml.go:2515-2584
        } else {
            const float s1 = elem(rl + 1);                // M is even and the block starts on an even row: the partner exists
            if (EPI == EPI_SILU_MUL) {                    // Silu(w1 h) * (w3 h): ml.go:2587-2589, 1877-1914 (llama.go:354-361)
                a.y[(size_t)c * a.ldy + (row >> 1)] = __fmul_rn(silu_ref(s0), s1);
            } else {                                      // RoPE on Q and the new K rows, K/V appended (llama.go:274-297)
                const uint32_t d = a.d, pos = a.past + c, half = a.hd >> 1;
                if (row < 2 * d) {
                    const uint32_t ee = row < d ? row : row - d;
                    const double2 cs = rope_s[c * half + ((ee % a.hd) >> 1)];
                    float o0, o1;
                    rope_rotate(s0, s1, cs, &o0, &o1);
                    float* dst = row < d ? a.q_out + (size_t)c * d + ee : a.k_cache + (size_t)pos * d + ee;
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

}  // namespace lh
