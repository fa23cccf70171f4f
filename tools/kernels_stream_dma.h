// tools/kernels_stream_dma.h — PROBE (not part of the product library): the weight-streaming GEMM of csrc/kernels_stream.h with loader
// waves that move their chunks global -> LDS directly (`buffer_load_dwordx4 ... lds`), for tools/stream_mm_check mode 4.
//
// Why: from 17 rows on k_stream_mm2 takes the SUM of its HBM time and its matrix-pipe time per chunk instead of
// their maximum, because the loader wave of a SIMD only issues in the gaps of the MFMA wave it shares the SIMD with (round-2 probes:
// one vector / memory instruction per 100-220 clocks next to back-to-back MFMAs).  k_stream_mm2's loader needs per chunk and wave one
// load + one ds_write_b128 per 1 KB plus the waits in between, and stages two chunks in registers.  Here a loader wave issues ONE
// instruction per 1 KB and nothing else: no staging registers, no LDS writes, no vector ALU (the lane's byte offset is a constant
// VGPR, the chunk offset an SGPR), and the chunks in flight live in a ring of NIMG LDS images instead of registers.
//   * image = [(MAXT + NCT) * 16 rows][KC floats], DENSE (an LDS-DMA instruction writes lane i at base + 16 i, so no row padding);
//     bank conflicts of the operand reads are avoided by a source-side swizzle as in k_gemm_glds: 16-byte granule g of image row r is
//     stored at granule position g ^ (r & 15).  An MFMA lane (row r16, slot) reading granule 4 kb + slot of 16 different rows then
//     touches 16 different positions.
//   * ring: chunk c lives in image c % NIMG.  Loader: wait until its own DMAs of chunk c have landed (vmcnt), barrier c, then request
//     chunk c + NIMG - 1 into the image chunk c - 1 has just left.  MFMA waves: barrier c, operands of chunk c, MFMAs.  One workgroup
//     barrier per chunk as in k_stream_mm2.
//   * MFMA side and summation structure as in k_stream_mm2 (k-blocks of a chunk dealt to the four MFMA waves, partial tiles added in
//     wave order): the results must equal k_stream_mm2's bit for bit.
// Plain epilogue only (store, one matrix): this file exists to measure the loop.
//
// STATE: written at the end of round 3 without GPU time left - compiled for gfx950 and its ISA read (loader loop: one
// `buffer_load_dwordx4 ... offen lds` per KB and scalar bookkeeping, no vector ALU; MFMA loop: 8 ds_read_b128 + 48 MFMAs per chunk for
// <6,2,64>), NOT yet run.  First thing to run next round (w1|w3-shaped launch, 32 / 48 / 64 rows; mode 2 = k_stream_mm2, mode 4 = this):
//   tools/build_probes.sh
//   for n in 32 48 64; do for m in 2 4; do tools/stream_mm_check 22016 4096 $n 64 $m; done; tools/stream_mm_check 22016 4096 $n 128 2; done
//   STREAM_DMA_IMAGES=4 tools/stream_mm_check 22016 4096 32 64 4      # deeper ring; 12288 4096 n = the wq|wk|wv shape, 4096 11008 n = w2
// The checker prints the error map against a float64 host product for every mode.
#pragma once
#include "../llama.go_amd/csrc/kernels_stream.h"

namespace lh {

__host__ __device__ inline size_t stream_dma_lds_bytes(int maxt, int nct, int kc, int nimg) { return (size_t)nimg * (maxt + nct) * 16 * kc * 4; }

template <int N>
__device__ __forceinline__ void wait_vm_dma() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));   // vmcnt(N); lgkmcnt / expcnt untouched
}

template <int MAXT, int NCT, int KC, int NIMG>
__global__ __launch_bounds__(2 * ST_TH) void k_stream_dma(const StreamArgs a) {
    static_assert(KC == 64 || KC == 128, "chunk");
    static_assert(NIMG >= 2 && NIMG <= 4, "ring");
    constexpr int GR = KC / 4;                  // 16-byte granules per image row
    constexpr int RPI = 64 / GR;                // image rows one DMA instruction covers (1 KB): 4 at KC = 64, 2 at KC = 128
    constexpr int ROWS = (MAXT + NCT) * 16;
    constexpr int NIW = ROWS / RPI / 4;         // DMA instructions per loader wave and chunk
    static_assert(ROWS % (RPI * 4) == 0, "rows per loader wave");
    constexpr int WAITN = NIW * (NIMG - 2) < 64 ? NIW * (NIMG - 2) : 63;   // (the counter holds 63: a stricter wait is still a correct one)
    constexpr size_t IMGF = (size_t)ROWS * KC;  // floats per image
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* img = (float*)smem_raw;              // [NIMG][ROWS][KC]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t T = a.M >> 4;
    const uint32_t t0 = (uint32_t)(((uint64_t)blockIdx.x * T) / gridDim.x), t1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * T) / gridDim.x);
    if (t1 <= t0) return;
    const uint32_t nt = t1 - t0;                // <= MAXT (host)
    const uint32_t nch = a.K / KC;
    const uint32_t r16 = (uint32_t)lane & 15, slot = (uint32_t)lane >> 4;
    constexpr int KB = KC / 64;                 // k-blocks (of 16 columns) per MFMA wave and chunk
    f4m acc[MAXT][NCT];
    if (wave < 4) {
        // ---- loader waves: instruction j of wave w covers image rows [(4 j + w) RPI, +RPI)
        const __amdgpu_buffer_rsrc_t rw = stream_rsrc(a.w[0]), rx = stream_rsrc(a.x);
        uint32_t voff[NIW];
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const uint32_t q = (uint32_t)j * 4 + (uint32_t)wave;
            const uint32_t rr = q * RPI + (uint32_t)lane / GR;          // image row this lane feeds
            const uint32_t gd = (uint32_t)lane % GR;                    // granule position it lands on
            const uint32_t gs = gd ^ (rr & 15u);                        // source granule stored there
            if (rr < (uint32_t)MAXT * 16) {
                uint32_t ti = rr >> 4;
                ti = ti < nt ? ti : nt - 1;                             // tiles beyond the block: a valid row, its sums are never stored
                voff[j] = (((t0 + ti) * 16 + (rr & 15u)) * a.K + gs * 4u) * 4u;
            } else {
                uint32_t c = rr - (uint32_t)MAXT * 16;
                c = c < a.n ? c : a.n - 1;
                voff[j] = (c * a.ldx + gs * 4u) * 4u;
            }
        }
        auto issue = [&](uint32_t ch) {
            const uint32_t cc = ch < nch ? ch : nch - 1;                // past the end: a harmless reload into a free image (uniform counts)
            const uint32_t k0b = cc * (uint32_t)KC * 4u;
            float* im = img + (size_t)(ch % NIMG) * IMGF;
#pragma unroll
            for (int j = 0; j < NIW; ++j) {
                const uint32_t q = (uint32_t)j * 4 + (uint32_t)wave;
                __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(im + (size_t)q * 256);   // 1 KB per instruction
                if (q * RPI < (uint32_t)MAXT * 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, dst, 16, (int)voff[j], (int)k0b, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, dst, 16, (int)voff[j], (int)k0b, 0, 0);
            }
        };
#pragma unroll
        for (int c = 0; c < NIMG - 1; ++c) issue((uint32_t)c);
        for (uint32_t ch = 0; ch < nch; ++ch) {
            wait_vm_dma<WAITN>();               // chunk ch of this wave has landed; the NIMG - 2 younger ones may still be in flight
            __builtin_amdgcn_s_barrier();       // barrier ch: every part of chunk ch is in its image, and the MFMA waves have left chunk ch - 1's
            issue(ch + NIMG - 1);               // ... whose image takes chunk ch + NIMG - 1
        }
        wait_vm_dma<0>();                       // the clamped tail requests
    } else {
        // ---- MFMA waves (k_stream_mm2's structure; operands out of the dense, swizzled image)
        const int cw = wave - 4;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[t][c] = f4m{0.f, 0.f, 0.f, 0.f};
        for (uint32_t ch = 0; ch < nch; ++ch) {
            const float* im = img + (size_t)(ch % NIMG) * IMGF;
            barrier_lds_only();                 // barrier ch (this wave's operand reads of chunk ch - 1 are complete: lgkmcnt(0))
            f4 af[KB][MAXT], bf[KB][NCT];
#pragma unroll
            for (int h = 0; h < KB; ++h) {
                const uint32_t g = ((uint32_t)(KB * cw + h) * 4 + slot) ^ r16;   // granule position of (k-block, slot) in a row with r & 15 = r16
#pragma unroll
                for (int c = 0; c < NCT; ++c) bf[h][c] = *(const f4*)(im + ((size_t)(MAXT * 16 + c * 16 + r16) * GR + g) * 4);
#pragma unroll
                for (int t = 0; t < MAXT; ++t) af[h][t] = *(const f4*)(im + ((size_t)(t * 16 + r16) * GR + g) * 4);
            }
            // k_stream_mm2's order (s outermost, then the wave's k-blocks) so that the sums round identically
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int h = 0; h < KB; ++h)
#pragma unroll
                    for (int t = 0; t < MAXT; ++t)
#pragma unroll
                        for (int c = 0; c < NCT; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[h][t][s], bf[h][c][s], acc[t][c], 0, 0, 0);
        }
    }
    __syncthreads();   // the images are dead
    // ---- the four MFMA waves' partial tiles meet in LDS and are added in wave order (k_stream_mm2's epilogue, plain store)
    constexpr int NC = NCT * 16;
    float* part = (float*)smem_raw;
    constexpr uint32_t TILE_FLOATS = 4u * NC * 16;
    const uint32_t batch = (uint32_t)(NIMG * IMGF / TILE_FLOATS);
    const uint32_t col = (uint32_t)tid >> 2, quad = (uint32_t)tid & 3;
    for (uint32_t tb = 0; tb < nt; tb += batch) {
        if (wave >= 4) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
                if ((uint32_t)t >= tb && (uint32_t)t < tb + batch && (uint32_t)t < nt) {
#pragma unroll
                    for (int c = 0; c < NCT; ++c)
                        *(f4m*)(part + (size_t)(t - tb) * TILE_FLOATS + ((size_t)(wave - 4) * NC + c * 16 + r16) * 16 + slot * 4) = acc[t][c];
                }
        }
        __syncthreads();
        if (tid < 4 * NC && col < a.n) {
            for (uint32_t t = tb; t < tb + batch && t < nt; ++t) {
                const float* p = part + (size_t)(t - tb) * TILE_FLOATS + (size_t)col * 16 + quad * 4;
                f4 s = *(const f4*)p;
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const f4 q = *(const f4*)(p + (size_t)w * NC * 16);
                    s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
                }
                *(f4*)(a.y[0] + (size_t)col * a.ldy + (t0 + t) * 16 + quad * 4) = s;
            }
        }
        __syncthreads();
    }
}

}  // namespace lh
