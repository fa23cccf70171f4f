// csrc/kernels_skinny.h — weight-streaming GEMM for SHORT prompts (2..16 token rows): one pass over the weights, fused like decode.
//
// Reference: server.Do feeds the whole prompt as ONE Eval (pkg/server/server.go:185-192), so time to first token for a short prompt
// is one pass over the weights: HBM-bound like decode (26.4 GB at 7B), not MFMA-bound.  Round 1 ran N <= 8 as five separate
// `gemm_small_n` launches per layer with norm / RoPE / SiLU as their own kernels, w2 streamed twice, and the 8-column register tile
// of k_gemv_cols spent more VALU time on its U*NC wave reductions than on FMAs (1.98 TB/s; 10.0 ms for 8 tokens vs a 3.3 ms floor).
//
// Here the contraction runs on the matrix cores purely as a REDUCTION ENGINE (MFMA only for dense W x contractions):
//   v_mfma_f32_16x16x4_f32   D[16 weight rows][16 token columns] += A[16 rows][4 k] * B[4 k][16 columns]
// accumulates over k inside the accumulator, so there is no cross-lane reduction at all.  At HBM rate the matrix pipe is ~1/3 busy.
//   - grid = #CU workgroups of 4 waves; each workgroup owns a contiguous block of weight rows (the same split as k_gemv), walked in
//     tiles of 16 rows; the 4 waves of a workgroup take interleaved 32-float k-blocks of the same tile (a row's 512 contiguous bytes
//     are requested by the 4 waves together), partial tiles meet in LDS once per tile (one barrier per 256 KB of weights);
//   - weights go global -> registers with non-temporal 16-byte loads, lane l = (row l % 16, k-group l / 16) holds exactly the A
//     operands of 8 consecutive MFMAs; a ring of RING k-blocks per wave stays in flight ACROSS tile boundaries;
//   - the activation rows live in LDS (<= 8 x 4096 floats), staged once per launch with the RMSNorm*gamma prologue applied on the way
//     (ml.go:1753-1812, 1877-1914); contractions longer than the LDS tile run as several launches over K-chunks with raw partial
//     sums handed through HBM (deterministic: sequential launches, fixed order);
//   - epilogues as in decode: residual add, SiLU*mul on (w1, w3) row pairs, RoPE + K/V cache append on (q, k, v) rows.
// Summation order differs from the scalar reference (k interleaved by 4, four partial sums): within 1e-4 like every MFMA path.
#pragma once
#include "kernels_llama.h"

namespace lh {

typedef float f4v __attribute__((ext_vector_type(4)));

struct SkinnyArgs {
    const float* w[3];      // matrix bases (MAP_BLOCK: [wq,wk,wv]; MAP_PAIR: [w1,w3])
    uint32_t rows_per_mat;  // MAP_BLOCK
    uint32_t M;             // virtual rows
    uint32_t K;             // full contraction length (row stride of the matrices, in floats)
    uint32_t k0, kc;        // this launch contracts columns [k0, k0 + kc), kc % 128 == 0
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
};

constexpr int SK_TH = 256, SK_NW = 4, SK_RING = 6, SK_KB = 32;

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
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t kc = a.kc, xstride = kc + 4;          // floats per staged activation row (+16 B: spreads the columns over the banks)
    float* xs = (float*)smem_raw;                        // [NP][xstride]
    float* red = xs + (size_t)NP * xstride;              // [2][SK_NW][64 * 4] partial tiles
    double* sred = (double*)(red + 2 * SK_NW * 256);     // [SK_NW] norm reduction
    const uint32_t nwg = gridDim.x, npairs = a.M >> 1;
    const uint32_t r0 = 2u * (uint32_t)(((uint64_t)blockIdx.x * npairs) / nwg);
    const uint32_t r1 = (blockIdx.x + 1 == nwg) ? a.M : 2u * (uint32_t)(((uint64_t)(blockIdx.x + 1) * npairs) / nwg);
    const uint32_t nrows = r1 - r0, ntiles = (nrows + 15) / 16;
    const uint32_t nkbw = kc / (SK_KB * SK_NW);          // k-blocks per wave per tile
    const uint32_t total = ntiles * nkbw;                // (tile, k-block) items of this wave, tile-major
    const uint32_t lm = lane & 15, lg = lane >> 4;       // lane = (row within tile / token column, k-group)

    // ---- weight ring: item i = (tile i / nkbw, k-block wave + 4 * (i % nkbw)); lane loads 8 consecutive floats of its row.
    // The load position (lt, lj) runs SK_RING items ahead of the compute position; the lane's row pointer is rebuilt once per tile.
    auto tile_ptr = [&](uint32_t t) -> const float* {
        uint32_t row = r0 + t * 16 + lm;
        row = row < r1 ? row : r1 - 1;                   // partial last tile: duplicates of the last row, dropped in the epilogue
        return skinny_row<MAP>(a, row) + a.k0 + (size_t)wave * SK_KB + lg * 8;
    };
    uint32_t lt = 0, lj = 0;
    const float* lp = tile_ptr(0);
    f4 wr[SK_RING][2];
    auto issue = [&](f4 (&slot)[2]) {                     // past the end: re-reads the last item (cache hit), never consumed
        typedef const f4 __attribute__((address_space(1))) gf4;   // rebuilt from integers: say GLOBAL, or the loads become flat_load
        gf4* p = (gf4*)(uintptr_t)(lp + (size_t)lj * (SK_KB * SK_NW));
        slot[0] = __builtin_nontemporal_load(p);
        slot[1] = __builtin_nontemporal_load(p + 1);
        if (lt < ntiles && ++lj == nkbw) {
            lj = 0;
            ++lt;
            if (lt < ntiles) lp = tile_ptr(lt); else lj = nkbw - 1;
        }
    };
#pragma unroll
    for (int s = 0; s < SK_RING; ++s) issue(wr[s]);

    // ---- stage the activation rows (whole workgroup), RMSNorm * gamma on the way
    for (uint32_t c = 0; c < (uint32_t)NP; ++c) {
        float* dst = xs + (size_t)c * xstride;
        if (c >= a.n) {                                  // unused columns: zeros (their results are never read)
            for (uint32_t k = tid; k < kc; k += SK_TH) dst[k] = 0.f;
            continue;
        }
        const float* xr = a.x + (size_t)c * a.ldx;
        float scale = 1.f;
        if (PRO == PRO_RMSNORM) {                        // over the FULL row, whatever this launch's chunk is
            double s = 0.0;
            for (uint32_t k = tid * 4; k < a.K; k += SK_TH * 4) {
                const f4 v = *(const f4*)(xr + k);
                s += (double)__fmul_rn(v.x, v.x); s += (double)__fmul_rn(v.y, v.y); s += (double)__fmul_rn(v.z, v.z); s += (double)__fmul_rn(v.w, v.w);
            }
            s = wave_sum_f64(s);
            __syncthreads();                             // sred reuse across columns
            if (lane == 0) sred[wave] = s;
            __syncthreads();
            const double mean = (((sred[0] + sred[1]) + sred[2]) + sred[3]) / (double)a.K;
            scale = (float)(1.0 / sqrt(mean + 1e-5));
        }
        for (uint32_t k = tid * 4; k < kc; k += SK_TH * 4) {
            f4 v = *(const f4*)(xr + a.k0 + k);
            if (PRO == PRO_RMSNORM) {
                const f4 g = *(const f4*)(a.gamma + a.k0 + k);
                v.x = __fmul_rn(g.x, __fmul_rn(v.x, scale)); v.y = __fmul_rn(g.y, __fmul_rn(v.y, scale));
                v.z = __fmul_rn(g.z, __fmul_rn(v.z, scale)); v.w = __fmul_rn(g.w, __fmul_rn(v.w, scale));
            }
            *(f4*)(dst + k) = v;
        }
    }
    __syncthreads();

    // ---- main stream
    const float* xl = xs + (size_t)(lm % NP) * xstride + lg * 8;   // this lane's B operands: column lm (mod NP), k-group lg
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    uint32_t tile = 0, j = 0;                            // position of item i
    for (uint32_t i0 = 0; i0 < total; i0 += SK_RING) {
#pragma unroll
        for (int s = 0; s < SK_RING; ++s) {
            const uint32_t i = i0 + s;
            if (i < total) {                             // wave-uniform
                const f4 w0 = wr[s][0], w1 = wr[s][1];
                issue(wr[s]);
                const float* xb = xl + (size_t)(wave + SK_NW * j) * SK_KB;
                const f4 x0 = *(const f4*)xb, x1 = *(const f4*)(xb + 4);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.x, x0.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.y, x0.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.z, x0.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w0.w, x0.w, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.x, x1.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.y, x1.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.z, x1.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w1.w, x1.w, acc, 0, 0, 0);
                if (++j == nkbw) {
                    // ---- tile done: the four K-quarters meet in LDS; thread (m, c) finishes element (row m, column c)
                    j = 0;
                    float* rb = red + (size_t)(tile & 1) * (SK_NW * 256);
                    *(f4v*)(rb + wave * 256 + lane * 4) = acc;      // lane holds D[4 * lg + i][lm], i = 0..3
                    acc = f4v{0.f, 0.f, 0.f, 0.f};
                    __syncthreads();
                    const uint32_t m = tid >> 4, c = tid & 15;
                    const uint32_t row = r0 + tile * 16 + m;
                    const bool pair_epi = (EPI == EPI_SILU_MUL || EPI == EPI_QKV_ROPE) && !a.part_out;
                    if (row < r1 && c < a.n && !(pair_epi && (m & 1))) {
                        auto elem = [&](uint32_t mm) {
                            const float* e = rb + ((mm >> 2) * 16 + c) * 4 + (mm & 3);
                            float s = ((e[0] + e[256]) + e[512]) + e[768];
                            const uint32_t rr = r0 + tile * 16 + mm;
                            if (a.part_in) s += a.part_in[(size_t)c * a.M + rr];
                            return s;
                        };
                        const float s0 = elem(m);
                        if (a.part_out) {
                            a.part_out[(size_t)c * a.M + row] = s0;
                        } else if (EPI == EPI_STORE) {
                            a.y[(size_t)c * a.ldy + row] = s0;
                        } else if (EPI == EPI_RESID) {
                            a.y[(size_t)c * a.ldy + row] = __fadd_rn(s0, a.resid[(size_t)c * a.ldy + row]);   // Add ml.go:2515-2584
                        } else {
                            const float s1 = elem(m + 1);
                            if (EPI == EPI_SILU_MUL) {           // Silu(w1 h) * (w3 h): ml.go:2587-2589, 1877-1914 (llama.go:354-361)
                                a.y[(size_t)c * a.ldy + (row >> 1)] = __fmul_rn(silu_ref(s0), s1);
                            } else {                             // RoPE on Q and the new K rows, K/V appended (llama.go:274-297)
                                const uint32_t d = a.d, pos = a.past + c;
                                if (row < 2 * d) {
                                    const uint32_t e = row < d ? row : row - d;
                                    const double2 cs = a.rope[(size_t)pos * (a.hd >> 1) + ((e % a.hd) >> 1)];
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
                    ++tile;
                }
            }
        }
    }
}

}  // namespace lh
