// csrc/kernels_q8.h — block-int8 weights (BASELINE.json configs[3]).
//
// The reference has NO quantised storage or kernels: ml.go only carries enum/size-table entries (ml.go:85-94, 123-124,
// QK = 32 at ml.go:24) and the loader rejects every non-f32/f16 dtype (llama.go:956-959).  The format is therefore ours,
// in the style of that ggml vintage's Q4_0 ({scale; QK quants} per block):
//     interchange / registration format: blocks of QK = 32 weights { float d; int8 q[32] } = 36 bytes,  w = d * q
//     quantiser (ours):                  d = max|w| / 127 (fp32 divide), q = clamp(rint(w / d), -127, 127)   (d = 0 -> q = 0)
// Semantics the checker uses: dequantise to fp32 (one rounding: fl32(d*q)), then the fp32 MulMat of ml.go:1976-2098.
//
// HBM layout (memory laid out for the GPU, not for the file): two planes per matrix — int8 quants [rows][K] and fp32
// scales [rows][K/32] — so a lane's 16-byte load is 16 consecutive weights, naturally aligned (36-byte blocks are not).
// Algorithmic bytes per weight stay 36/32.
//
// GEMV: same fat-workgroup weight stream as the fp32 kernel, but a row is 4x fewer bytes, so the 1024 threads form
// G = 1024/TPR row groups that stream G rows at once (same bytes in flight per CU).  Thread t of a group owns columns
// 16t..16t+15 (+16*TPR*j); its 16 activations per chunk stay in registers.  Per chunk: 16 cvt + 16 fma in fp32, then
// one fma with the block scale (the scale is factored out of the block sum; vs dequantise-first this moves one rounding,
// ~1e-7 relative).
#pragma once
#include "kernels_llama.h"

namespace lh {

__device__ __forceinline__ u4 ld_nt_u4(const u4* p) { return __builtin_nontemporal_load(p); }


// 16 int8 x 16 fp32: two independent accumulator PAIRS so the FMAs can issue as v_pk_fma_f32 (2 FMAs per instruction)
__device__ __forceinline__ float dot16_q8(const u4 q, const f4 (&x)[4]) {
    f2 a0 = {0.f, 0.f}, a1 = {0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int d = (int)q[k];
        const f2 w0 = {(float)(int)(signed char)(d), (float)(int)(signed char)(d >> 8)};
        const f2 w1 = {(float)(int)(signed char)(d >> 16), (float)(d >> 24)};
        a0 = __builtin_elementwise_fma(w0, f2{x[k].x, x[k].y}, a0);
        a1 = __builtin_elementwise_fma(w1, f2{x[k].z, x[k].w}, a1);
    }
    return (a0.x + a0.y) + (a1.x + a1.y);
}

// ---------------------------------------------------------------------------------------------------
// k_gemv_q8s — the same stream with SCALAR row addressing and an explicit two-set software pipeline.
// What the first kernel's ISA showed (round 2): because `a.w[v & 1]` / `a.ws[m]` index the kernel-argument arrays with a per-lane
// value, every batch began with global_load_dwordx2 of the matrix POINTERS, an s_waitcnt that also drained the weight loads still in
// flight (vmcnt counts in order), ~40 VALU instructions of 64-bit address arithmetic and selects, and six v_mov to rotate the
// register sets: only one batch was ever in flight per wave and the kernel sat at ~5 TB/s with the VALU half idle.  Here:
//   - a row group is a whole number of waves (TPR >= 64), so its row index is made wave-uniform with readfirstlane; matrix bases are
//     hoisted into scalar registers; a load is `global_load_dwordx4 v, v_off, s[row]` with a constant per-lane byte offset;
//   - rows beyond the workgroup's range are redirected (scalar select) to the cache-resident activation vector instead of branching;
//   - the loop is unrolled over two register sets (A consumed while B is in flight, then B while A), so no register is copied.
// Arithmetic and summation order are those of k_gemv_q8: results are bit-identical.
// ---------------------------------------------------------------------------------------------------
template <int KI, int U, int TPR, int PRO, int EPI, int MAP, int TH = 1024>
__global__ __launch_bounds__(TH) void k_gemv_q8s(const GemvArgs a) {
    LH_TOUCH_ARGS(a.w[0], a.x, a.hd, a.wg_r);   // the argument block's lines behind one wait (kernels_common.h)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int G = TH / TPR, NWR = TPR / 64;
    static_assert(TPR % 64 == 0, "a row group must be a whole number of waves");
    double* sred = (double*)smem_raw;            // [16]
    float* red = (float*)(smem_raw + 16 * 8);    // [rows of this workgroup][NWR]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tr = tid % TPR, wr = wave % NWR;
    const uint32_t grp = (uint32_t)__builtin_amdgcn_readfirstlane(tid / TPR);   // uniform within a wave
    const uint32_t K = a.K, K16 = K >> 4;
    uint32_t r0, r1;
    wg_row_block(a.M, a.wg_q, a.wg_r, &r0, &r1);
    // Matrix bases as scalar integers: base of matrix 0 plus the DISTANCES to matrices 1 and 2, so that choosing a matrix is
    // `base + (m >= 1 ? d1 : 0) + (m == 2 ? d2 : 0)` — selects against the constant 0.  (A select among three pointer variables is
    // folded by the compiler into an indexed read of a table it builds in scratch memory, which vectorises the whole address path:
    // seen in the ISA, 2x slower.)
    const uint64_t q0 = (uint64_t)sgpr_ptr(a.w[0]), s0 = (uint64_t)sgpr_ptr(a.ws[0]);
    const uint64_t dq1 = MAP == MAP_SINGLE ? 0 : (uint64_t)sgpr_ptr(a.w[1]) - q0, ds1 = MAP == MAP_SINGLE ? 0 : (uint64_t)sgpr_ptr(a.ws[1]) - s0;
    const uint64_t dq2 = MAP == MAP_BLOCK ? (uint64_t)sgpr_ptr(a.w[2]) - q0 - dq1 : 0, ds2 = MAP == MAP_BLOCK ? (uint64_t)sgpr_ptr(a.ws[2]) - s0 - ds1 : 0;
    const uint64_t xdummy = (uint64_t)sgpr_ptr(a.x);   // 4K bytes: covers a quant row (K bytes) and a scale row (K/8 bytes)
    const uint32_t rpm = a.rows_per_mat;
    // MAP_BLOCK (round 6): "virtual" bases base_m - m * rpm * K, so that a row's address is  virtual base + row * K  with the VIRTUAL row index, and
    // the workgroup's matrix is chosen once: its rows cross at most one matrix boundary (`bnd`), which costs one compare + select per row.  The
    // general form (m from two compares, r = row - m * rpm, three selects per plane) stood in front of every row's loads, and with one wave per
    // SIMD nothing hides scalar work: wq|wk|wv of 7B 12.7 -> 11.8 us standalone, against 10.9 us for the same bytes as ONE matrix and 32.9 / 32.7 us
    // on the fp32 kernel, whose rows are four times as long (tools/q8s_phase_probe, profiles/r06_q8s_phase_probe.txt).
    const uint64_t srow = (uint64_t)(K >> 5) * 4u;
    // (selects against the constant 0 again: a select among three 64-bit values becomes a table in scratch memory, 4x slower - measured)
    const uint64_t e1q = dq1 - (uint64_t)rpm * K, e2q = dq2 - (uint64_t)rpm * K, e1s = ds1 - (uint64_t)rpm * srow, e2s = ds2 - (uint64_t)rpm * srow;
    const uint32_t mlo = (r0 >= rpm ? 1u : 0u) + (r0 >= 2u * rpm ? 1u : 0u), mhi = (r1 - 1u >= rpm ? 1u : 0u) + (r1 - 1u >= 2u * rpm ? 1u : 0u);
    const bool one_mat = mhi - mlo <= 1u;     // at most ONE matrix boundary inside the workgroup's rows (always, unless a matrix is shorter than a workgroup's block)
    const uint64_t qv_wg = q0 + (mlo >= 1u ? e1q : 0) + (mlo == 2u ? e2q : 0), sv_wg = s0 + (mlo >= 1u ? e1s : 0) + (mlo == 2u ? e2s : 0);
    const uint32_t bnd = (mlo + 1u) * rpm;    // rows from here on belong to the next matrix
    const uint64_t nq = (mlo == 0u ? e1q : 0) + (mlo == 1u ? e2q : 0), ns = (mlo == 0u ? e1s : 0) + (mlo == 1u ? e2s : 0);
    // Epilogue operands first of all: for wq|wk|wv they hang on a dependent chain (position -> RoPE entry) whose scalar load then flies while the x / gamma
    // loads are issued; behind the x loads the wave waited for it in front of its first weight row (+0.45 us on that launch), and behind the first weight
    // rows the chain waits for the rows (loads return in order: 11.97 -> 12.72 us).  tools/q8s_phase_probe.
    const uint32_t fin = (EPI == EPI_STORE || EPI == EPI_RESID) ? (uint32_t)tid : 2u * (uint32_t)tid;
    float resid_pre;
    double2 cs_pre;
    uint32_t past_pre;
    gemv_prefetch_fin<EPI>(a, r0, r1, fin, &resid_pre, &cs_pre, &past_pre);
    f4 xr[KI][4];
    bool act[KI];
    uint32_t qoff[KI], soff[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        const uint32_t c = tr + j * TPR;
        act[j] = c < K16;
        qoff[j] = act[j] ? c * 16u : 0u;
        soff[j] = act[j] ? (c >> 1) * 4u : 0u;
#pragma unroll
        for (int k = 0; k < 4; ++k) xr[j][k] = act[j] ? ((const f4*)a.x)[c * 4 + k] : f4{0.f, 0.f, 0.f, 0.f};
    }
    // norm weights requested together with x (the first kernel fetched them AFTER the norm's barrier: one more L2 round trip on the
    // critical path of a 15-25 us kernel)
    f4 gr[PRO == PRO_RMSNORM ? KI : 1][4];
    if (PRO == PRO_RMSNORM) {
#pragma unroll
        for (int j = 0; j < KI; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) gr[j][k] = act[j] ? ((const f4*)a.gamma)[(tr + j * TPR) * 4 + k] : f4{0.f, 0.f, 0.f, 0.f};
    }

    // slot u of this group holds row rb + G*u
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
            // addresses rebuilt from integers: say that they are GLOBAL memory, or the loads become flat_load (counted on both wait counters)
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
    constexpr uint32_t STEP = G * U;

    if (PRO == PRO_RMSNORM) {
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
                    const f4 g = gr[j][k];
                    xr[j][k].x = __fmul_rn(g.x, __fmul_rn(xr[j][k].x, scale));
                    xr[j][k].y = __fmul_rn(g.y, __fmul_rn(xr[j][k].y, scale));
                    xr[j][k].z = __fmul_rn(g.z, __fmul_rn(xr[j][k].z, scale));
                    xr[j][k].w = __fmul_rn(g.w, __fmul_rn(xr[j][k].w, scale));
                }
            }
    }

    auto consume = [&](const u4 (&wd)[U][KI], const float (&sd)[U][KI], uint32_t rb) {
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KI; ++j) s = fmaf(sd[u][j], dot16_q8(wd[u][j], xr[j]), s);
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
    };
    // (both register sets requested in front of the norm: nothing, 18.60 / 18.58 us on w1|w3, 12.73 / 12.99 on wq|wk|wv - same probe)
    for (uint32_t rb = r0 + grp; rb < r1; rb += 2 * STEP) {
        fetch(wB, scB, rb + STEP);
        consume(wA, scA, rb);
        fetch(wA, scA, rb + 2 * STEP);
        consume(wB, scB, rb + STEP);   // rows >= r1: loaded from the dummy, results dropped by the row test
    }
    __syncthreads();
    gemv_finish<EPI, NWR>(a, red, r0, r1, fin, resid_pre, cs_pre, past_pre);
}

}  // namespace lh
