// csrc/kernels_gemm.h — fp32 MFMA GEMM for prefill (N >= 32 tokens per Eval):
//     Y[n][m] (+ R[n][m]) = sum_k X[n][k] * W[m][k]          (both operands K-contiguous: MulMat's "NT" shape, ml.go:295-318)
// gfx950 has an exact-f32 matrix instruction, v_mfma_f32_32x32x2_f32 (D = A*B + C as a k-ordered fmaf chain, no
// reduced-precision path exists on CDNA4), at the f32 vector rate (157 TF peak) but with far fewer issue slots and
// operand registers than a VALU GEMM.  Prefill is the compute-bound side of the hot path: 13B, N = 1024 is 27 TFLOP
// against 51 GB of weights, so this kernel is priced against the MFMA roof, not HBM.
//
// Tiling: workgroup = 256 threads (4 waves, 2 x 2) computes a 128(n) x 128(m) tile; each wave owns 64 x 64 = 2 x 2 MFMA
// tiles (4 accumulators x 16 VGPRs).  K advances in slabs of BK = 32 through LDS, stored K-MAJOR ([k][row], leading
// dimension 129) so that the MFMA operand fetch — lane l needs A[i = l & 31][k = l >> 5] — is one conflict-free
// ds_read_b32 per operand (lanes 0-31: 32 consecutive rows of one k; lanes 32-63: the next k, other lane group).
// The next slab is fetched from global memory into registers (coalesced float4, 8 rows x 128 B per wave-instruction)
// while the current one is multiplied, and written to the second LDS buffer: one barrier per slab.
// blockIdx -> tile mapping keeps the 8 workgroups that share a weight panel on one XCD (L2 reuse of W across the
// N dimension), using the observed block -> XCD round-robin (speed only, never correctness).
#pragma once
#include "kernels_common.h"

namespace lh {

typedef float f16v __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float* x;  // [N][K] rows at ldx
    const float* w;  // [M][K]
    float* y;        // [N][M] rows at ldy
    const float* r;  // optional residual, same layout as y
    uint32_t N, M, K, ldx, ldy;
};

constexpr int GBM = 128, GBN = 128, GBK = 32, GLD = 129;

__global__ __launch_bounds__(256) void k_gemm_mfma(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                       // [2][GBK][GLD]
    float* Ws = smem + 2 * GBK * GLD;       // [2][GBK][GLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;  // wave position inside the 128 x 128 tile
    // tile mapping: tiles_m panels of W; consecutive groups of 8 n-tiles of the same panel share an XCD
    const uint32_t tiles_n = (a.N + GBN - 1) / GBN, tiles_m = (a.M + GBM - 1) / GBM;
    uint32_t bid = blockIdx.x;
    {
        const uint32_t nx = 8, total = tiles_n * tiles_m;
        if (total % nx == 0) {  // bijective XCD remap: blocks with equal (bid % 8) land on one XCD
            const uint32_t per = total / nx;
            bid = (bid % nx) * per + bid / nx;
        }
    }
    const uint32_t tm = bid / tiles_n, tn = bid % tiles_n;  // n fastest: neighbours share the W panel
    const uint32_t n0 = tn * GBN, m0 = tm * GBM;

    // global -> register staging: thread covers rows (tid/8 + 32 p), k = 4*(tid%8)..+3
    const int lr = tid >> 3, lk = (tid & 7) * 4;
    f4 xg[4], wg[4];
    auto fetch = [&](uint32_t k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const uint32_t n = n0 + lr + 32 * p, m = m0 + lr + 32 * p;
            xg[p] = n < a.N ? *(const f4*)(a.x + (size_t)n * a.ldx + k0 + lk) : f4{0.f, 0.f, 0.f, 0.f};
            wg[p] = m < a.M ? *(const f4*)(a.w + (size_t)m * a.K + k0 + lk) : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto stash = [&](int buf) {
        float* xs = Xs + buf * GBK * GLD;
        float* ws = Ws + buf * GBK * GLD;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = lr + 32 * p;
            xs[(lk + 0) * GLD + row] = xg[p].x; xs[(lk + 1) * GLD + row] = xg[p].y;
            xs[(lk + 2) * GLD + row] = xg[p].z; xs[(lk + 3) * GLD + row] = xg[p].w;
            ws[(lk + 0) * GLD + row] = wg[p].x; ws[(lk + 1) * GLD + row] = wg[p].y;
            ws[(lk + 2) * GLD + row] = wg[p].z; ws[(lk + 3) * GLD + row] = wg[p].w;
        }
    };

    f16v acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
        const float* xs = Xs + buf * GBK * GLD + wn * 64 + li;
        const float* ws = Ws + buf * GBK * GLD + wm * 64 + li;
#pragma unroll
        for (int ks = 0; ks < GBK; ks += 2) {
            const float a0 = xs[(ks + lh) * GLD], a1 = xs[(ks + lh) * GLD + 32];
            const float b0 = ws[(ks + lh) * GLD], b1 = ws[(ks + lh) * GLD + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) stash(buf ^ 1);
        __syncthreads();
    }
    // C/D layout (dtype-independent on gfx950): col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t m = m0 + wm * 64 + j * 32 + li;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t n = n0 + wn * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (n < a.N && m < a.M) {
                    const size_t o = (size_t)n * a.ldy + m;
                    float v = acc[i][j][e];
                    if (a.r) v = __fadd_rn(v, a.r[o]);
                    a.y[o] = v;
                }
            }
        }
}

}  // namespace lh
