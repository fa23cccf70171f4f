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
    const float* w[3];  // per group [M][K]
    float* y[3];        // per group [N][M] rows at ldy
    const float* r[3];  // per group optional residual, same layout as y
    uint32_t groups, N, M, K, ldx, ldy;
    uint32_t ldw;                 // row pitch of W in floats (0 = K)
    uint64_t xbs, wbs, ybs;       // blockIdx.y batch strides in floats (attention: one batch entry per head)
};

constexpr int GBK = 32;

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

    const uint32_t nk = a.K / GBK;
    fetch(0);
    stash(0);
    __syncthreads();
    const int li = lane & 31, lh = lane >> 5;
    for (uint32_t kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * GBK);
        const float* xs = Xs + buf * GBK * LDX + wn * TN * 32 + li;
        const float* ws = Ws + buf * GBK * LDW + wm * TM * 32 + li;
#pragma unroll
        for (int ks = 0; ks < GBK; ks += 2) {
            float af[TN], bf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) af[i] = xs[(ks + lh) * LDX + 32 * i];
#pragma unroll
            for (int j = 0; j < TM; ++j) bf[j] = ws[(ks + lh) * LDW + 32 * j];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) stash(buf ^ 1);
        __syncthreads();
    }
    // C/D layout (dtype-independent on gfx950): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const uint32_t m = m0 + (wm * TM + j) * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t n = n0 + (wn * TN + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (n < a.N && m < a.M) {
                    const size_t o = (size_t)n * a.ldy + m;
                    float v = acc[i][j][e];
                    if (R) v = __fadd_rn(v, R[o]);
                    Y[o] = v;
                }
            }
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
