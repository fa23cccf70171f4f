// tools/kernels_stream_eq.h — NOT IN THE PRODUCT (round 5: built, checked, slower than k_stream_mm2 on every 7B launch; profiles/r05_stream_eq_probe.txt).
// fp32 weights, ONE column tile (2..16 token rows: a short prompt, or the pods of a tick), sixteen equal waves.
//
// Reference: ComputeForwardMulMatFP32 (pkg/ml/ml.go:1976-2098); with `gamma` the RMSNorm * weight in front of it (ml.go:1753-1812, 1877-1914).
//
// Why (round 5): the 9..16-row launches were the last ones on round 2's k_stream_mm2 (loader waves moving global -> registers -> LDS with
// ds_write_b128, 13 LDS-port cycles per KB): 16 pods ran at 0.587 of the HBM roofline.  The LDS-DMA kernel of round 4 (k_stream_dma) had
// measured slower here only because it cannot fold the RMSNorm (its loader waves move raw bytes), which cost a launch per norm.  What the
// block-int8 work of this round showed carries over (kernels_stream_q8b.h):
//   * sixteen equal waves: every wave issues a few of the chunk's LDS-DMA instructions, then works on its share - the vector ALU of a
//     SIMD serves four waves' instructions in the time one wave issues one, so the norm's multiplies (gamma on the operand-read side) and
//     sums of squares cost next to nothing; the DMA is inline asm (a builtin makes the compiler drain the ring in front of every operand read);
//   * the norm folds the way k_stream_mm2 folded it: W is contracted with gamma * x and the sums are multiplied by the per-token scale
//     s = fl32(1 / sqrt(mean(x^2) + 1e-5)) in the epilogue; the squares are accumulated by the waves that read the activations anyway
//     (fp32 squares, four of them meet in fp32, the groups add up in f64, fixed order).
// Structure: chunks of KC = 64 columns; image = weights [MAXT * 16 rows][64 floats] + activations [16 rows][64 floats] (granule g of row r
// at g ^ (r & 15), k_stream_dma's layout) + gamma [64 floats]; ring of NIMG images, one workgroup barrier per chunk.  Wave w multiplies
// k-block w & 3 (16 columns = four v_mfma_f32_16x16x4_f32 per tile) of the tiles w >> 2, (w >> 2) + 4: the four waves of a SIMD
// (w, w + 4, w + 8, w + 12) hold the same k-block of all tiles, so the matrix pipes are evenly loaded whatever the tile count.
// Partial tiles of the four k-block waves meet in LDS in wave order (stream_epilogue); fused epilogues, tile pairs and batched rows as in
// the other stream kernels.
#pragma once
#include "../llama.go_amd/csrc/kernels_stream.h"

namespace lh {

constexpr int SEQ_TH = 1024, SEQ_KC = 64;
__host__ __device__ constexpr size_t stream_eq_image_bytes(int maxt) { return (size_t)(maxt * 16 + 16) * SEQ_KC * 4 + SEQ_KC * 4; }
__host__ __device__ constexpr size_t stream_eq_lds_bytes(int maxt, int nimg) { return (size_t)nimg * stream_eq_image_bytes(maxt) + 1024; }   // + the norm's partial sums and scales

template <int MAXT, int NIMG>
__global__ __launch_bounds__(SEQ_TH) void k_stream_eq(const StreamArgs a) {
    static_assert(NIMG >= 2 && NIMG <= 5, "ring");
    constexpr int KC = SEQ_KC, NWV = SEQ_TH / 64;
    constexpr int NB = KC / 16;                 // k-blocks per chunk: 4
    constexpr int TG = NWV / NB;                // tile groups: 4
    constexpr int TPW = (MAXT + TG - 1) / TG;   // tiles per wave
    constexpr int GR = KC / 4;                  // 16-byte granules per image row: 16
    constexpr int RPI = 64 / GR;                // image rows per DMA instruction: 4
    constexpr int NWI = MAXT * 16 / RPI, NXI = 16 / RPI, NI = NWI + NXI + 1;   // weights, activations, gamma
    constexpr int NIW = (NI + NWV - 1) / NWV;
    constexpr int WAITN = NIW * (NIMG - 2) < 64 ? NIW * (NIMG - 2) : 63;
    constexpr uint32_t W_BYTES = MAXT * 16 * KC * 4, X_BYTES = 16 * KC * 4, G_BYTES = KC * 4, IMG_BYTES = W_BYTES + X_BYTES + G_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tiles_per_mat = a.M >> 4, T = tiles_per_mat * a.groups;
    const bool pairs = a.epi == ST_EPI_SILU_MUL;   // virtual tile v = (tile v >> 1 of matrix v & 1), dealt in PAIRS (k_stream_mm2)
    const uint32_t units = pairs ? tiles_per_mat : T, um = pairs ? 2u : 1u;
    const uint32_t t0 = um * (uint32_t)(((uint64_t)blockIdx.x * units) / gridDim.x), t1 = um * (uint32_t)(((uint64_t)(blockIdx.x + 1) * units) / gridDim.x);
    if (t1 <= t0) return;
    const uint32_t nt = t1 - t0;
    const uint32_t nch = a.K / KC;
    const bool norm = a.gamma != nullptr;
    const uint32_t r16 = (uint32_t)lane & 15, slot = (uint32_t)lane >> 4;
    // ---- this wave's DMA instructions of a chunk: piece q = NWV j + wave (weight rows, activation rows, gamma)
    const char* base[NIW];
    uint32_t voff[NIW], doff[NIW];
    bool small[NIW];   // the gamma piece: 16 active lanes
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        uint32_t q = (uint32_t)j * NWV + (uint32_t)wave;
        q = q < (uint32_t)NI ? q : (uint32_t)NI - 1;                       // surplus slots repeat the last piece (same bytes to the same place)
        const uint32_t rr = q * RPI + (uint32_t)lane / GR;                 // image row this lane feeds
        const uint32_t gd = (uint32_t)lane % GR, gs = gd ^ (rr & 15u);     // granule position it lands on, source granule stored there
        small[j] = q == (uint32_t)(NWI + NXI);
        if (q < (uint32_t)NWI) {
            uint32_t ts = (q * RPI) >> 4;
            ts = ts < nt ? ts : nt - 1;                                    // slots beyond the block: a valid tile again, its sums are never stored
            const uint32_t v = t0 + ts;
            uint32_t g, tile;
            if (pairs) { g = v & 1u; tile = v >> 1; }
            else { g = v / tiles_per_mat; tile = v - g * tiles_per_mat; }
            base[j] = (const char*)((g == 0 ? a.w[0] : (g == 1 ? a.w[1] : a.w[2])) + (size_t)tile * 16 * a.K);
            voff[j] = ((rr & 15u) * a.K + gs * 4u) * 4u;
            doff[j] = q * 1024u;
        } else if (q < (uint32_t)(NWI + NXI)) {
            uint32_t c = rr - (uint32_t)MAXT * 16;
            c = c < a.n ? c : a.n - 1;                                     // rows past the batch: the last row again (never stored)
            base[j] = (const char*)a.x;
            voff[j] = (c * a.ldx + gs * 4u) * 4u;
            doff[j] = q * 1024u;
        } else {
            base[j] = (const char*)(norm ? a.gamma : a.x);
            voff[j] = ((uint32_t)lane & 15u) * 16u;
            doff[j] = W_BYTES + X_BYTES;
        }
    }
    // (inline asm on purpose: kernels_stream_q8b.h - through the builtin the compiler drains every DMA in flight in front of the next operand read)
    typedef int i4v __attribute__((ext_vector_type(4)));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    auto issue = [&](uint32_t ch) {
        const uint32_t im = lds0 + (ch % NIMG) * IMG_BYTES;
        const uint32_t so = (uint32_t)__builtin_amdgcn_readfirstlane((int)(ch * KC * 4u));
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const uint64_t b = (uint64_t)sgpr_ptr(base[j]);
            const i4v rs = {(int)(uint32_t)b, (int)((uint32_t)(b >> 32) & 0xffffu), 0x7fffffff, 0x00020000};   // raw buffer, stride 0 (stream_rsrc's words)
            const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(im + doff[j]));
            if (small[j]) {
                unsigned long long saved;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %5\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(m0v), "v"(voff[j]), "s"(rs), "s"(so), "s"(0xffffull) : "memory", "m0");
            } else if (j * NWV + wave < NWI) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" :: "s"(m0v), "v"(voff[j]), "s"(rs), "s"(so) : "memory", "m0");   // weights: read once
            } else {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(voff[j]), "s"(rs), "s"(so) : "memory", "m0");
            }
        }
    };
#pragma unroll
    for (int c = 0; c < NIMG - 1; ++c) if ((uint32_t)c < nch) issue((uint32_t)c);
    // ---- this wave's share of a chunk: k-block kb of the tiles tg, tg + TG
    const uint32_t kb = (uint32_t)wave % NB, tg = (uint32_t)wave / NB;
    f4m acc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) acc[j] = f4m{0.f, 0.f, 0.f, 0.f};
    const uint32_t gq = ((kb * 4 + slot) ^ r16) * 16;                      // byte offset of (k-block, slot) in an image row with r & 15 = r16
    uint32_t woff[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        uint32_t t = tg + (uint32_t)j * TG;
        t = t < (uint32_t)MAXT ? t : (uint32_t)MAXT - 1;                   // (a slot past the tiles: the last one again, its sums are never stored)
        woff[j] = (t * 16 + r16) * (KC * 4) + gq;
    }
    const uint32_t xoff = W_BYTES + r16 * (KC * 4) + gq, goff = W_BYTES + X_BYTES + (kb * 4 + slot) * 16;
    double ssq = 0.0;
    for (uint32_t ch = 0; ch < nch; ++ch) {
        const char* im = smem_raw + (size_t)(ch % NIMG) * IMG_BYTES;
        {
            const uint32_t left = nch - 1 - ch;                            // younger chunks that exist: min(NIMG - 2, left) of them stay in flight
            if (left >= (uint32_t)(NIMG - 2)) wait_vm<WAITN>();
            else if (NIMG >= 4 && left >= 1) { if (left == 1) wait_vm<(NIW < 64 ? NIW : 63)>(); else wait_vm<(2 * NIW < 64 ? 2 * NIW : 63)>(); }
            else wait_vm<0>();
        }
        barrier_lds_only();                     // barrier ch: chunk ch is in its image; everybody has left chunk ch - 1's ...
        if (ch + NIMG - 1 < nch) issue(ch + NIMG - 1);   // ... which takes chunk ch + NIMG - 1
        f4 bf = *(const f4*)(im + xoff);
        f4 af[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) af[j] = *(const f4*)(im + woff[j]);
        if (norm) {
            const f4 gv = *(const f4*)(im + goff);
            if (tg == 0) ssq += (double)__fadd_rn(__fadd_rn(__fmul_rn(bf.x, bf.x), __fmul_rn(bf.y, bf.y)), __fadd_rn(__fmul_rn(bf.z, bf.z), __fmul_rn(bf.w, bf.w)));
            bf.x = __fmul_rn(gv.x, bf.x); bf.y = __fmul_rn(gv.y, bf.y); bf.z = __fmul_rn(gv.z, bf.z); bf.w = __fmul_rn(gv.w, bf.w);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < TPW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j][s], bf[s], acc[j], 0, 0, 0);
    }
    // ---- the folded norm's per-token scales: the k-block waves of tile group 0 hold the squares of their columns
    float* scl = (float*)(smem_raw + (size_t)NIMG * IMG_BYTES);            // [16] scales, then [NB][16] partial sums (f64) behind them
    double* psum = (double*)(scl + 16);
    if (norm) {
        if (tg == 0) {
            ssq += __shfl_xor(ssq, 16, 64);
            ssq += __shfl_xor(ssq, 32, 64);
            if (lane < 16) psum[kb * 16 + lane] = ssq;
        }
        __syncthreads();
        if (tid < 16) {
            double t = psum[tid];
#pragma unroll
            for (int w = 1; w < NB; ++w) t += psum[w * 16 + tid];
            scl[tid] = (float)(1.0 / sqrt(t / (double)a.K + 1e-5));
        }
    }
    __syncthreads();   // the images are dead (every wave waited for its DMAs); the scales are written
    stream_epilogue<MAXT, 1, 1, false, NB>(a, smem_raw, (uint32_t)((size_t)NIMG * IMG_BYTES / 4), norm ? (const float*)scl : nullptr, t0, nt, 0, tiles_per_mat,
                                            [&](int t, int) { return acc[t / TG]; });
}

}  // namespace lh
