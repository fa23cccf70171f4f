// csrc/kernels_gemm.h — fp32 MFMA GEMM for prefill (N >= 32 tokens per Eval):
//     Y_g[n][m] (+ R_g[n][m]) = sum_k X[n][k] * W_g[m][k]      g = 0..groups-1 (wq|wk|wv or w1|w3 share X and one launch)
// (both operands K-contiguous: MulMat's "NT" shape, ml.go:295-318).  gfx950 has an exact-f32 matrix instruction,
// v_mfma_f32_32x32x2_f32 (D = A*B + C as a k-ordered fmaf chain; no reduced-precision f32 path exists on CDNA4), at the f32
// vector rate (157 TF peak) with far fewer issue slots and operand registers than a VALU GEMM.  Prefill is the
// compute-bound side of the hot path (13B, N = 1024: 27 TFLOP against 51 GB of weights), priced against the MFMA roof.
//
// Tiling: 256 threads = 4 waves arranged WN x WM; each wave owns TN x TM MFMA tiles of 32 x 32 (TN*TM accumulators of
// 16 VGPRs); workgroup tile = (WN*TN*32) tokens x (WM*TM*32) weight rows.  Three shapes are instantiated and the host picks
// the one with the least tile-count quantisation for the launch (a 128 x 128 tile on a 1024 x 5120 output is 320 tiles on 256
// CUs = 62 % balance; 128 x 160 is exactly one tile per CU):
//     <2,2,2,2> 128 x 128     <2,2,2,1> 128 x 64     <4,1,1,5> 128 x 160
// K advances in slabs of 32 through LDS stored K-MAJOR ([k][row], leading dimension = rows + 1) so the MFMA operand fetch
// — lane l needs A[i = l & 31][k = l >> 5] — is one conflict-free ds_read_b32 (lanes 0-31: 32 consecutive rows of one k;
// lanes 32-63: the next k, other lane group).  The next slab is fetched from global memory into registers (coalesced
// float4, 8 rows x 128 B per wave-instruction) while the current one is multiplied, then written to the other LDS buffer:
// one barrier per slab.  blockIdx -> tile mapping keeps the n-tiles that share a weight panel on one XCD (L2 reuse of W),
// using the observed block -> XCD round-robin (speed only, never correctness).
#pragma once
#include "kernels_common.h"

namespace lh {

typedef float f16v __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* x;     // [N][K] rows at ldx
    const float* w[3];  // per group [M][K] (k_gemm_q8: the int8 quant plane, [M][K] bytes)
    const float* ws[3]; // k_gemm_q8: per group block scales [M][K/32]
    float* y[3];        // per group [N][M] rows at ldy
    const float* r[3];  // per group optional residual, same layout as y
    uint32_t groups, N, M, K, ldx, ldy;
    uint32_t ldw;                 // row pitch of W in floats (0 = K)
    uint64_t xbs, wbs, ybs;       // batch strides in floats (attention: one batch entry per head)
    uint32_t batch;               // k_gemm_glds: number of batch entries (0 = 1); k_gemm_mfma takes the batch from gridDim.y
    // Causal structure of prefill attention (row n = query at position past + n, DiagMaskInf ml.go:2377-2414): the reference
    // computes the full block and masks afterwards; the masked part contributes exact zeros, so it can be skipped.
    //   causal = 1 (scores S = Q.K^T, columns = keys): a tile whose first key lies beyond its last query's position is not
    //              computed (softmax rewrites those entries with 0 without reading them);
    //   causal = 2 (O = P.V, contraction over keys): P[n][t] = 0 for t > past + n, so the K loop stops after the last key any
    //              row of the tile can see.
    uint32_t causal, past;
    // Split-K (k_gemm_glds, few tiles: short prompts).  splits > 1: work item = (tile, K range ks); the partial product goes to
    // part[group][ks][N][M] (row pitch M, no residual) and k_splitk_reduce adds the partials in ks order (+ residual) into Y:
    // two passes, fixed order, bit-reproducible — no atomics.
    uint32_t splits;
    float* part;
    // Fused epilogues of k_gemm_glds (long prompts; the launches llama.Eval makes, llama.go:255-297 and :346-361):
    //   GEMM_EPI_SILU_MUL  groups = 1, M = 2 F VIRTUAL rows: row v = row v >> 1 of w[v & 1] (w1, w3); y[0][n][v >> 1] = silu(w1 h) * (w3 h)
    //   GEMM_EPI_QKV_ROPE  groups = 3 (wq, wk, wv), M = d: RoPE on Q and the new K rows, K / V rows appended to the cache at past + n
    uint32_t epi;
    float* q_out;
    float* k_cache;
    float* v_cache;
    const double2* rope;
    uint32_t hd;
    // k_gemm_b9 (kernels_gemm_b9.h): X as three bf16 planes (x = hi + mid + lo exactly; row n of plane p at xs + p * xs_plane + n * ldxs, elements
    // of 2 bytes); x above is then unused
    const uint16_t* xs;
    uint64_t xs_plane;
    uint32_t ldxs;
#if defined(GEMM_CLOCK) || defined(B9_TRACE)
    unsigned long long* clk;   // tools/gemm_probe.hip -DGEMM_CLOCK: shader clocks and 100 MHz ticks one workgroup spent in k_gemm_glds
#endif
};
enum { GEMM_EPI_STORE = 0, GEMM_EPI_SILU_MUL = 1, GEMM_EPI_QKV_ROPE = 2 };

constexpr int GBK = 32;
#ifndef LH_GST
#define LH_GST 3
#endif
constexpr int GST = LH_GST;  // LDS stages of k_gemm_glds (tools/gemm_probe.hip may override)

// C/D layout (dtype-independent on gfx950): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
// The residual is fetched for a whole 32 x 32 block before any of it is used, through clamped (always valid) addresses: a load
// inside `if (n < N && m < M)` waits out a memory latency per element (80 dependent round trips per lane, ~40 us per tile).
template <int TN, int TM>
__device__ __forceinline__ void gemm_store(const GemmArgs& a, f16v (&acc)[TN][TM], float* Y, const float* R, uint32_t nb, uint32_t mb, int li, int lh,
                                           uint32_t ldy) {
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const uint32_t m = mb + j * 32 + li;
            const uint32_t mc = m < a.M ? m : a.M - 1;
            float rv[16];
            if (R) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const uint32_t n = nb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    rv[e] = R[(size_t)(n < a.N ? n : a.N - 1) * ldy + mc];
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t n = nb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (n < a.N && m < a.M) Y[(size_t)n * ldy + m] = R ? __fadd_rn(acc[i][j][e], rv[e]) : acc[i][j][e];
            }
        }
}


// Fused epilogues (GemmArgs::epi).  Partner rows (w1 / w3 of one ff row; the two rows of a RoPE pair) are adjacent D columns = adjacent
// LANES (col = lane & 31), so the partner's value comes by one cross-lane exchange per element; every lane runs the exchange.
template <int TN, int TM>
__device__ __forceinline__ void gemm_store_fused(const GemmArgs& a, f16v (&acc)[TN][TM], uint32_t g, uint32_t nb, uint32_t mb, int li, int lh) {
    const bool even = (li & 1) == 0;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const uint32_t m = mb + j * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t n = nb + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const float v = acc[i][j][e];
                const float p = __shfl_xor(v, 1, 64);
                const bool ok = n < a.N && m < a.M;
                if (a.epi == GEMM_EPI_SILU_MUL) {
                    // even lane: w1 row m >> 1, its partner: w3 row m >> 1.  Silu then Mul: ml.go:2587-2589, 1877-1914 (llama.go:354-361)
                    if (ok && even) a.y[0][(size_t)n * a.ldy + (m >> 1)] = __fmul_rn(silu_ref(v), p);
                } else {
                    const uint32_t pos = a.past + n;
                    if (g < 2) {   // Rope mode 0 on Q / mode 1 on the new K rows (ml.go:2253-2328)
                        const double2 cs = a.rope[(size_t)(ok ? pos : a.past) * (a.hd >> 1) + (((m < a.M ? m : 0) % a.hd) >> 1)];
                        float o0, o1;
                        rope_rotate(even ? v : p, even ? p : v, cs, &o0, &o1);
                        if (ok) (g == 0 ? a.q_out + (size_t)n * a.M : a.k_cache + (size_t)pos * a.M)[m] = even ? o0 : o1;
                    } else if (ok) {
                        a.v_cache[(size_t)pos * a.M + m] = v;   // cache append llama.go:274-278
                    }
                }
            }
        }
}

template <int WN, int WM, int TN, int TM>
__global__ __launch_bounds__(256) void k_gemm_mfma(const GemmArgs a) {
    static_assert(WN * WM == 4, "4 waves");
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32, LDX = BN + 1, LDW = BM + 1;
    constexpr int PX = BN / 32, PW = BM / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                      // [2][GBK][LDX]
    float* Ws = smem + 2 * GBK * LDX;      // [2][GBK][LDW]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    const uint32_t tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const uint32_t per_group = tiles_n * tiles_m, total = per_group * a.groups;
    uint32_t bid = blockIdx.x;
    if (total % 8 == 0) {  // bijective XCD remap: blocks with equal (bid % 8) land on one XCD and get consecutive tiles
        const uint32_t per = total / 8;
        bid = (bid % 8) * per + bid / 8;
    }
    const uint32_t g = bid / per_group, t = bid % per_group;
    const uint32_t tm = t / tiles_n, tn = t % tiles_n;  // n fastest: neighbours share the W panel
    const uint32_t n0 = tn * BN, m0 = tm * BM;
    if (a.causal == 1 && m0 > a.past + n0 + BN - 1) return;
    const uint32_t ldw = a.ldw ? a.ldw : a.K;
    const float* X = a.x + (size_t)blockIdx.y * a.xbs;
    const float* W = a.w[g] + (size_t)blockIdx.y * a.wbs;
    float* Y = a.y[g] + (size_t)blockIdx.y * a.ybs;
    const float* R = a.r[g] ? a.r[g] + (size_t)blockIdx.y * a.ybs : nullptr;

    // global -> register staging: thread covers rows (tid/8 + 32 p), k = 4*(tid%8)..+3
    const int lr = tid >> 3, lk = (tid & 7) * 4;
    f4 xg[PX], wg[PW];
    auto fetch = [&](uint32_t k0) {
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const uint32_t n = n0 + lr + 32 * p;
            xg[p] = n < a.N ? *(const f4*)(X + (size_t)n * a.ldx + k0 + lk) : f4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const uint32_t m = m0 + lr + 32 * p;
            wg[p] = m < a.M ? *(const f4*)(W + (size_t)m * ldw + k0 + lk) : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
        float* xs = Xs + buf * GBK * LDX;
        float* ws = Ws + buf * GBK * LDW;
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int row = lr + 32 * p;
            xs[(lk + 0) * LDX + row] = xg[p].x; xs[(lk + 1) * LDX + row] = xg[p].y;
            xs[(lk + 2) * LDX + row] = xg[p].z; xs[(lk + 3) * LDX + row] = xg[p].w;
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int row = lr + 32 * p;
            ws[(lk + 0) * LDW + row] = wg[p].x; ws[(lk + 1) * LDW + row] = wg[p].y;
            ws[(lk + 2) * LDW + row] = wg[p].z; ws[(lk + 3) * LDW + row] = wg[p].w;
        }
    };

    f16v acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#ifndef GEMM_ABL
#define GEMM_ABL 0   // tools/gemm_probe.hip: 1 = no global fetch, 2 = + no LDS stash, 3 = + no barrier, 4 = + operands not re-read from LDS
#endif
    uint32_t nk = a.K / GBK;
    if (a.causal == 2) nk = min(nk, (a.past + n0 + BN + GBK - 1) / GBK);
    fetch(0);
    stash(0);
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5;
    for (uint32_t kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (GEMM_ABL < 1 && kt + 1 < nk) fetch((kt + 1) * GBK);
        const float* xs = Xs + buf * GBK * LDX + wn * TN * 32 + li;
        const float* ws = Ws + buf * GBK * LDW + wm * TM * 32 + li;
#pragma unroll
        for (int ks = 0; ks < GBK; ks += 2) {
            float af[TN], bf[TM];
            const int kr = GEMM_ABL >= 4 ? 0 : ks;
#pragma unroll
            for (int i = 0; i < TN; ++i) af[i] = xs[(kr + lh) * LDX + 32 * i];
#pragma unroll
            for (int j = 0; j < TM; ++j) bf[j] = ws[(kr + lh) * LDW + 32 * j];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (GEMM_ABL < 2 && kt + 1 < nk) stash(buf ^ 1);
        if (GEMM_ABL < 3) __syncthreads();
    }
    gemm_store<TN, TM>(a, acc, Y, R, n0 + wn * TN * 32, m0 + wm * TM * 32, li, lh, a.ldy);
}

// ---- the same GEMM with LDS-DMA operand staging --------------------------------------------------------------------
// tools/gemm_probe (13B prefill shapes) prices the register-staged kernel above at 70-75 % of the 155.7 TFLOP/s this chip
// sustains on bare v_mfma_f32_32x32x2_f32: global loads into staging registers cost 6-14 points, the 4-byte transposing LDS
// stores 5-7, the 4-byte operand reads 8; the barrier itself is free.  This variant removes all three:
//   * operands travel global -> LDS by global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass).  The LDS image is
//     row-major, 32 floats = 128 B per tile row, 8 rows = 1 KB per wave-instruction.  LDS-DMA writes lane-linearly, so the
//     bank swizzle is applied on the SOURCE side: lane L of a piece (row L>>3, slot L&7) fetches 16-byte granule
//     (L&7) ^ key(row) of its row — still one full 128-byte line per 8 lanes.
//   * an MFMA step may pair any two k's as long as A and B agree, so lane (row i, half h) takes one ds_read_b128 = granule
//     2q+h of its row and feeds its 4 floats to 4 consecutive MFMAs (step e multiplies k = 8q+e and k = 8q+4+e): 4x fewer LDS
//     instructions, conflict-free through the XOR on the slot (slot = granule ^ key(row), key = (row >> 1) & 7).
//   * a ring of GST LDS stages; once per K-slab: counted vmcnt (MY pieces of the next slab landed, newer ones stay in
//     flight), raw s_barrier (everybody's landed), DMA of slab kt+GST into the stage just drained.
// Rows past N / M load a clamped address; their products land in accumulator entries the epilogue never stores.
// Accumulation order differs from the kernel above (k pairs re-grouped): a different fixed order of exact-f32 fmaf, the same
// class of difference as any other summation order (|diff| ~ 1e-7 relative).
template <int WN, int WM, int TN, int TM>
__global__ __launch_bounds__(256) void k_gemm_glds(const GemmArgs a) {
    static_assert(WN * WM == 4, "4 waves");
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32, ROWS = BN + BM, STAGE = ROWS * 32, PIECES = ROWS / 8, PPW = PIECES / 4;
    static_assert(PIECES % 4 == 0, "pieces divide over the 4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [GST][ROWS][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    const uint32_t tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const uint32_t splits = a.splits ? a.splits : 1;
    const uint32_t per_group = tiles_n * tiles_m, total = per_group * a.groups, work = total * (a.batch ? a.batch : 1) * splits;
    const uint32_t ldw = a.ldw ? a.ldw : a.K;
    const int li = lane & 31, lh = lane >> 5, sw = (li >> 1) & 7;
    const uint32_t nk_full = a.K / GBK;
    // Persistent workgroups, ONE per CU (the host pads the LDS request so that two cannot co-reside): with the software
    // pipeline below a lone workgroup keeps the matrix pipe busier (84-86 %) than two co-resident ones (81-83 %), and a grid of
    // perfectly balanced tiles otherwise ends with the last partial round packed two-per-CU onto half of the chip.
    // Work item w = batch entry * tiles + tile; workgroup b owns w = v, v + G, v + 2G, ... with v = (b % 8) * G/8 + b / 8:
    // the workgroups of one XCD (b % 8, observed round-robin; speed only) run CONSECUTIVE tiles, n fastest, so the tiles that
    // share a weight panel meet in one L2.
    const uint32_t G = gridDim.x;
    const uint32_t v0 = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
#ifdef GEMM_CLOCK
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (uint32_t wv = v0; wv < work; wv += G) {
    const uint32_t ks = wv % splits, wi = wv / splits;  // K range fastest: the pieces of one tile run side by side
    const uint32_t be = wi / total, bid = wi % total;
    const uint32_t g = bid / per_group, t = bid % per_group;
    const uint32_t tm = t / tiles_n;
    uint32_t tn = t % tiles_n;
    // causal P.V: a tile's contraction length grows with its row block, and with G % tiles_n == 0 the static assignment would hand
    // a workgroup the SAME row block every round (the unluckiest one the longest tile each time: 47 % balance at 8 row blocks).
    // Odd rounds take the row blocks in reverse, so consecutive rounds of a workgroup sum to the same length.
    if (a.causal == 2 && G % tiles_n == 0 && ((wv / G) & 1)) tn = tiles_n - 1 - tn;
    const uint32_t n0 = tn * BN, m0 = tm * BM;
    if (a.causal == 1 && m0 > a.past + n0 + BN - 1) continue;
    const uint32_t nk = splits > 1 ? nk_full / splits : a.causal == 2 ? min(nk_full, (a.past + n0 + BN + GBK - 1) / GBK) : nk_full;
    const uint32_t kbase = ks * nk * GBK;  // first column of this work item's K range (0 unless split)
    const float* X = a.x + (size_t)be * a.xbs + kbase;
    const float* W = a.w[g] + (size_t)be * a.wbs + kbase;
    float* Y = splits > 1 ? a.part + ((size_t)(g * splits + ks) * a.N) * a.M : a.y[g] + (size_t)be * a.ybs;
    const float* R = (splits > 1 || !a.r[g]) ? nullptr : a.r[g] + (size_t)be * a.ybs;
    const uint32_t ldy = splits > 1 ? a.M : a.ldy;
    __builtin_amdgcn_s_barrier();  // every wave is done reading the previous tile's stages

    // piece p (8 tile rows) is fetched by wave p % 4; this lane's source pointer for each of its pieces
    const float* src[PPW];
    // swizzle key of tile row r = (r >> 1) & 7: a 128-byte row covers HALF of the 64 LDS banks (even rows the lower half, odd
    // rows the upper), so the 8 rows of equal parity among 16 consecutive ones must land in 8 different granule slots
    // (with key = r & 7 SQ_LDS_BANK_CONFLICT was 50 % of the LDS-active cycles).  Row of lane L in piece P: 8 P + (L >> 3).
    const uint32_t gran = (uint32_t)((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7));
#pragma unroll
    for (int pp = 0; pp < PPW; ++pp) {
        const uint32_t row = (uint32_t)(wave + 4 * pp) * 8 + (uint32_t)(lane >> 3);  // tile row: X rows first, then W rows
        if (row < (uint32_t)BN) {
            const uint32_t n = n0 + row;
            src[pp] = X + (size_t)(n < a.N ? n : a.N - 1) * a.ldx + 4 * gran;
        } else {
            const uint32_t m = m0 + row - BN;
            const uint32_t mm = m < a.M ? m : a.M - 1;
            if (a.epi == GEMM_EPI_SILU_MUL)   // virtual row mm = row mm >> 1 of w1 (even) / w3 (odd); base + distance, never a pointer select
                src[pp] = (const float*)((uint64_t)a.w[0] + ((mm & 1u) ? (uint64_t)a.w[1] - (uint64_t)a.w[0] : 0)) + (size_t)(mm >> 1) * ldw + 4 * gran;
            else
                src[pp] = W + (size_t)mm * ldw + 4 * gran;
        }
    }
    auto issue = [&](int stage, uint32_t k0, int p0, int p1) {
#pragma unroll
        for (int pp = p0; pp < p1; ++pp)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(GEMM_ABL == 8 ? a.x + lane * 4 : src[pp] + k0),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + (wave + 4 * pp) * 256), 16, 0, 0);
    };

    f16v acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Software pipeline.  Inside a slab the operands of granule pair q+1 are read (second register set) in the MIDDLE of the
    // MFMAs of q, so neither the reads' latency nor their issue slots stall the matrix pipe (sched_group_barrier pins the
    // interleaving: mask 0x008 = MFMA, 0x100 = DS read; left alone hipcc puts each q's reads directly in front of their first
    // use).  The slab boundary sits in the middle of q = 3: by then every read of the current stage has completed (its data
    // feeds MFMAs already issued), so after "my pieces of the next slab landed" + s_barrier the wave may read q = 0 of the next
    // stage AND start the DMA of slab kt+2 into the current one, all under the second half of q = 3's MFMAs.
    constexpr int NM = 4 * TN * TM, NR = TN + TM;
    f4 af[2][TN], bf[2][TM];
    auto fetch_ops = [&](int set, int stage, int q) {
        const float* xs = smem + stage * STAGE + (wn * TN * 32 + li) * 32;
        const float* ws = smem + stage * STAGE + (BN + wm * TM * 32 + li) * 32;
        const int slot = ((2 * q + lh) ^ sw) * 4;
#pragma unroll
        for (int i = 0; i < TN; ++i) af[set][i] = *(const f4*)(xs + i * 32 * 32 + slot);
#pragma unroll
        for (int j = 0; j < TM; ++j) bf[set][j] = *(const f4*)(ws + j * 32 * 32 + slot);
    };
    auto mfmas = [&](int set, int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; ++e)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i][e], bf[set][j][e], acc[i][j], 0, 0, 0);
    };
    // GST stages in a ring, slab j in stage j % GST.  "Landed" = all but the newest GST-2 slabs of MY pieces have arrived
    // (vmcnt counts in order), then the barrier makes that true for everybody's.  Slabs past the end re-read the last one
    // (keeps the count uniform; harmless).
    auto landed = [&]() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (GST - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto slab_k0 = [&](uint32_t j) { return (j < nk ? j : nk - 1) * GBK; };
    // DMA schedule: slab kt+GST-1 is requested DURING slab kt, a third of its pieces after the first MFMA half of each of
    // q = 0, 1, 2, into the stage drained at the previous boundary.  (All PPW instructions back to back at the boundary cost
    // 7-8 % of the MFMA time: the wave cannot issue MFMAs while the texture path takes its 1-KB requests, and the matrix pipe
    // holds only the one in flight.)
    constexpr int P1 = (PPW + 2) / 3, P2 = (2 * PPW + 2) / 3;
#pragma unroll
    for (int j = 0; j < GST - 1; ++j) issue(j, slab_k0(j), 0, PPW);
    landed();
    fetch_ops(0, 0, 0);
    uint32_t st = 0, sp = GST - 1;  // kt % GST, (kt - 1) % GST
    for (uint32_t kt = 0; kt < nk; ++kt) {
        const uint32_t kn = slab_k0(kt + GST - 1);
#define LH_GEMM_QBLOCK(q, p0, p1)                                                                      \
        if (GEMM_ABL < 7 || GEMM_ABL >= 8) fetch_ops(((q) + 1) & 1, st, (q) + 1);                                       \
        if (GEMM_ABL < 5 || GEMM_ABL >= 8) issue(sp, kn, p0, p1);                                                       \
        mfmas((q) & 1, 0, 4);                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, NM / 2, 0);                                        \
        __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x020, (p1) - (p0), 0);                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, NM - NM / 2, 0);
        LH_GEMM_QBLOCK(0, 0, P1)
        LH_GEMM_QBLOCK(1, P1, P2)
        LH_GEMM_QBLOCK(2, P2, PPW)
#undef LH_GEMM_QBLOCK
        mfmas(1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            const uint32_t sn = st + 1 == GST ? 0 : st + 1;
            if (GEMM_ABL < 6 || GEMM_ABL == 8) landed();
            if (GEMM_ABL < 7 || GEMM_ABL >= 8) fetch_ops(0, sn, 0);
            sp = st;
            st = sn;
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 2, 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail DMA
    if (a.epi != GEMM_EPI_STORE) gemm_store_fused<TN, TM>(a, acc, g, n0 + wn * TN * 32, m0 + wm * TM * 32, li, lh);
    else gemm_store<TN, TM>(a, acc, Y, R, n0 + wn * TN * 32, m0 + wm * TM * 32, li, lh, ldy);
  }
#ifdef GEMM_CLOCK
    if (a.clk && blockIdx.x == G / 2 && tid == 0) { a.clk[0] = __builtin_amdgcn_s_memtime() - clk_c0; a.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0; }
#endif
}

// Second pass of split-K: Y_g[n][m] = (R_g[n][m] +) sum over ks, in ks order, of part[g][ks][n][m].
__global__ __launch_bounds__(256) void k_splitk_reduce(const GemmArgs a) {
    const uint32_t M4 = a.M / 4;
    const uint64_t per = (uint64_t)a.N * M4, tot = per * a.groups;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (uint64_t)gridDim.x * 256) {
        const uint32_t g = (uint32_t)(i / per);
        const uint64_t r = i % per;
        const uint32_t n = (uint32_t)(r / M4), m = (uint32_t)(r % M4) * 4;
        const float* p = a.part + ((size_t)(g * a.splits) * a.N + n) * a.M + m;
        f4 s = *(const f4*)p;
        for (uint32_t k = 1; k < a.splits; ++k) {
            const f4 v = *(const f4*)(p + (size_t)k * a.N * a.M);
            s.x = __fadd_rn(s.x, v.x); s.y = __fadd_rn(s.y, v.y); s.z = __fadd_rn(s.z, v.z); s.w = __fadd_rn(s.w, v.w);
        }
        const size_t o = (size_t)n * a.ldy + m;
        if (a.r[g]) {
            const f4 v = *(const f4*)(a.r[g] + o);
            s.x = __fadd_rn(s.x, v.x); s.y = __fadd_rn(s.y, v.y); s.z = __fadd_rn(s.z, v.z); s.w = __fadd_rn(s.w, v.w);
        }
        *(f4*)(a.y[g] + o) = s;
    }
}

// ---- block-int8 weights: dequantising GEMM (prefill of config-4 models) -------------------------------------------------
// Y_g = X . dequant(W_g)^T with W in the device layout of kernels_q8.h (int8 plane [M][K] + fp32 scales [M][K/32], one scale
// per 32 weights = per K-slab).  Semantics: w = fl32(d * q), then the exact-f32 MFMA chain — literally "dequantise, then the
// fp32 MulMat" (DESIGN 3b), no reordering of the scale.  X travels by LDS-DMA exactly as in k_gemm_glds; a W slab row is 32
// bytes, so each thread loads 16 quants + the row's scale into registers a slab ahead, converts (16 cvt + 16 mul) and stores
// the 64 bytes into the same swizzled row-major image the MFMA loop reads (4 x ds_write_b128).  Two LDS stages; everything
// for slab b+2 is issued at the boundary inside slab b (after the barrier that retires the reads of the stage it replaces).
template <int WN, int WM, int TN, int TM>
__global__ __launch_bounds__(256) void k_gemm_q8(const GemmArgs a) {
    static_assert(WN * WM == 4, "4 waves");
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32, ROWS = BN + BM, STAGE = ROWS * 32, XPIECES = BN / 8, XPW = XPIECES / 4, PW = (BM + 127) / 128;
    static_assert(XPIECES % 4 == 0, "X pieces divide over the 4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][ROWS][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    const uint32_t tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    const uint32_t splits = a.splits ? a.splits : 1;  // split-K exactly as in k_gemm_glds
    const uint32_t per_group = tiles_n * tiles_m, total = per_group * a.groups * splits;
    const int li = lane & 31, lh = lane >> 5, sw = (li >> 1) & 7;
    const uint32_t nk = a.K / GBK / splits, KB = a.K / 32;
    const uint32_t G = gridDim.x;
    const uint32_t v0 = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const uint32_t gran = (uint32_t)((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7));
    const int wrow = tid >> 1, whalf = tid & 1;  // W staging: thread -> (row within a 128-row pass, 16-weight half of the slab)
  for (uint32_t wv = v0; wv < total; wv += G) {
    const uint32_t ks = wv % splits, wi = wv / splits;
    const uint32_t g = wi / per_group, t = wi % per_group;
    const uint32_t tm = t / tiles_n, tn = t % tiles_n;
    const uint32_t n0 = tn * BN, m0 = tm * BM;
    const uint32_t kbase = ks * nk * GBK;  // first column of this work item's K range
    const float* X = a.x + kbase;
    const signed char* Wq = (const signed char*)a.w[g] + kbase;
    const float* Ws = a.ws[g] + kbase / 32;
    float* Y = splits > 1 ? a.part + ((size_t)(g * splits + ks) * a.N) * a.M : a.y[g];
    const float* R = splits > 1 ? nullptr : a.r[g];
    const uint32_t ldy = splits > 1 ? a.M : a.ldy;
    __builtin_amdgcn_s_barrier();  // every wave is done reading the previous tile's stages

    const float* src[XPW];
#pragma unroll
    for (int pp = 0; pp < XPW; ++pp) {
        const uint32_t n = n0 + (uint32_t)(wave + 4 * pp) * 8 + (uint32_t)(lane >> 3);
        src[pp] = X + (size_t)(n < a.N ? n : a.N - 1) * a.ldx + 4 * gran;
    }
    const signed char* wsrc[PW];
    const float* ssrc[PW];
#pragma unroll
    for (int ps = 0; ps < PW; ++ps) {
        const uint32_t m = m0 + ps * 128 + wrow;
        const uint32_t mc = m < a.M ? m : a.M - 1;
        wsrc[ps] = Wq + (size_t)mc * a.K + 16 * whalf;
        ssrc[ps] = Ws + (size_t)mc * KB;
    }
    auto issue_x = [&](int stage, uint32_t kt) {
#pragma unroll
        for (int pp = 0; pp < XPW; ++pp)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[pp] + kt * GBK),
                                             (__attribute__((address_space(3))) void*)(smem + stage * STAGE + (wave + 4 * pp) * 256), 16, 0, 0);
    };
    u4 wq[PW];
    float wd[PW];
    auto load_w = [&](uint32_t kt) {  // unconditional (clamped rows); rows past BM are never stored
#pragma unroll
        for (int ps = 0; ps < PW; ++ps) {
            wq[ps] = *(const u4*)(wsrc[ps] + (size_t)kt * 32);
            wd[ps] = ssrc[ps][kt];
        }
    };
    auto store_w = [&](int stage) {
#pragma unroll
        for (int ps = 0; ps < PW; ++ps) {
            const int row = ps * 128 + wrow;
            if (row < BM) {
                float* dst = smem + stage * STAGE + (BN + row) * 32;
                const int key = (row >> 1) & 7;  // BN is a multiple of 16: key(BN + row) = key(row)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = (int)wq[ps][i];
                    f4 f;
                    f.x = __fmul_rn(wd[ps], (float)(int)(signed char)(v));
                    f.y = __fmul_rn(wd[ps], (float)(int)(signed char)(v >> 8));
                    f.z = __fmul_rn(wd[ps], (float)(int)(signed char)(v >> 16));
                    f.w = __fmul_rn(wd[ps], (float)(v >> 24));
                    *(f4*)(dst + 4 * ((4 * whalf + i) ^ key)) = f;
                }
            }
        }
    };

    f16v acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    constexpr int NM = 4 * TN * TM, NR = TN + TM;
    f4 af[2][TN], bf[2][TM];
    auto fetch_ops = [&](int set, int stage, int q) {
        const float* xs = smem + stage * STAGE + (wn * TN * 32 + li) * 32;
        const float* ws = smem + stage * STAGE + (BN + wm * TM * 32 + li) * 32;
        const int slot = ((2 * q + lh) ^ sw) * 4;
#pragma unroll
        for (int i = 0; i < TN; ++i) af[set][i] = *(const f4*)(xs + i * 32 * 32 + slot);
#pragma unroll
        for (int j = 0; j < TM; ++j) bf[set][j] = *(const f4*)(ws + j * 32 * 32 + slot);
    };
    auto mfmas = [&](int set, int e0, int e1) {
#pragma unroll
        for (int e = e0; e < e1; ++e)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[set][i][e], bf[set][j][e], acc[i][j], 0, 0, 0);
    };
    auto landed = [&]() {  // my DMA pieces and my LDS stores are done; after the barrier so are everybody's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto slab = [&](uint32_t j) { return j < nk ? j : nk - 1; };  // past the end: re-read the last slab (never consumed)
    issue_x(0, 0);
    load_w(0);
    store_w(0);
    issue_x(1, slab(1));
    load_w(slab(1));
    store_w(1);
    load_w(slab(2));
    landed();
    fetch_ops(0, 0, 0);
    for (uint32_t kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            fetch_ops((q + 1) & 1, st, q + 1);
            mfmas(q & 1, 0, 4);
            __builtin_amdgcn_sched_group_barrier(0x008, NM / 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NR, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NM - NM / 2, 0);
        }
        mfmas(1, 0, 2);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            landed();                      // slab kt+1 is complete in stage st^1; every read of stage st has retired
            fetch_ops(0, st ^ 1, 0);
            store_w(st);                   // slab kt+2 (registers loaded one boundary ago) -> the stage just drained
            issue_x(st, slab(kt + 2));
            load_w(slab(kt + 3));
        }
        __builtin_amdgcn_sched_barrier(0);
        mfmas(1, 2, 4);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    gemm_store<TN, TM>(a, acc, Y, R, n0 + wn * TN * 32, m0 + wm * TM * 32, li, lh, ldy);
  }
}

// Scale + causal mask + softmax on the full score block S[h][j][0..Tp) in place (Scale ml.go:2331-2374, DiagMaskInf
// ml.go:2377-2414, SoftMax ml.go:2432-2505): row j keeps keys t <= past + j; masked and padding columns become exactly 0.
__global__ __launch_bounds__(256) void k_softmax_causal(float* __restrict__ S, uint32_t N, uint32_t Tp, uint32_t past, float scale) {
    __shared__ float scratch[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t j = blockIdx.x, h = blockIdx.y;
    float* p = S + ((size_t)h * N + j) * Tp;
    const uint32_t T = past + j + 1;
    float m = -INFINITY;
    for (uint32_t t = tid; t < T; t += 256) {
        const float v = __fmul_rn(p[t], scale);
        p[t] = v;
        m = fmaxf(m, v);
    }
    m = wave_max(m);
    if (lane == 0) scratch[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(scratch[0], scratch[1]), fmaxf(scratch[2], scratch[3]));
    __syncthreads();
    float s = 0.f;
    for (uint32_t t = tid; t < T; t += 256) {
        const float v = (float)exp((double)__fsub_rn(p[t], m));
        p[t] = v;
        s += v;
    }
    s = wave_sum(s);
    if (lane == 0) scratch[wave] = s;
    __syncthreads();
    const float inv = __fdiv_rn(1.0f, (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]));
    for (uint32_t t = tid; t < Tp; t += 256) p[t] = t < T ? __fmul_rn(p[t], inv) : 0.f;
}

// V cache rows [t][h*hd + c] -> VT[h][c][t] (t padded to Tp with zeros): the reference's VTrans copy (llama.go:315-322),
// needed because the MFMA GEMM wants both operands contiguous along the contraction (here: keys).
__global__ __launch_bounds__(256) void k_transpose_v(const float* __restrict__ v_cache, float* __restrict__ vt, uint32_t T, uint32_t Tp, uint32_t d, uint32_t hd) {
    __shared__ float tile[32][33];
    const uint32_t h = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const uint32_t t = t0 + r;
        tile[r][tx] = t < T ? v_cache[(size_t)t * d + h * hd + c0 + tx] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const uint32_t c = c0 + r, t = t0 + tx;
        if (t < Tp) vt[((size_t)h * hd + c) * Tp + t] = tile[tx][r];
    }
}

}  // namespace lh
