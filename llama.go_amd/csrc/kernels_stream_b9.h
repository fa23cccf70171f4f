// csrc/kernels_stream_b9.h — fp32 weights, 17..64 token rows per weight pass, on the bf16 matrix pipe with EXACT products (round 6).
//     Y_g[c][m] (+ R_g[c][m]) = sum_k X[c][k] * W_g[m][k]        (ComputeForwardMulMatFP32, pkg/ml/ml.go:1976-2098)
// The rows are a short prompt's tokens (server.Do feeds the prompt as ONE Eval, pkg/server/server.go:185-192) or the pods of a tick
// (server.go:84-106).
//
// Why: k_stream_dma (kernels_stream.h) multiplies on v_mfma_f32_16x16x4_f32, which issues at the fp32 VECTOR rate (32 clocks per SIMD for
// 16 x 16 x 4); from 33 rows on its launches are bound by the matrix pipe, not by the weight stream (64 rows: w1|w3 of 7B 110-130 us against a
// 53 us stream, matrix pipe 78 % busy), and the chip drops to 2.09 GHz beside the HBM stream.  An fp32 number is exactly the sum of three bf16
// (8 + 8 + 8 significand bits: split3, kernels_stream.h), so
//     x * w = (xh + xm + xl) * (wh + wm + wl) = nine products of 8-bit significands, each EXACT in fp32,
// i.e. nine v_mfma_f32_16x16x32_bf16 (9 x 16 clocks) contract what eight fp32 MFMAs (8 x 32 clocks) do, with no narrow-precision input
// anywhere: what differs from the fp32 instruction is the order in which exact products meet in the fp32 accumulator (small terms first), as
// any tiling changes it.  SURVEY App. C forbids LOSSY narrow inputs; this is the lossless split of k_stream_q8b (activations) and k_gemm_b9
// (both sides).
//
// Structure = k_stream_dma's (wave-specialised: four loader waves that only issue `buffer_load_dwordx4 ... lds`, four MFMA waves - one per
// SIMD - that never touch global memory; ring of NIMG LDS images, one workgroup barrier per 64-column chunk, the MFMA waves one chunk behind
// the images with their operands in registers).  Three builds with k_stream_q8b's sixteen EQUAL waves came first and lost
// (profiles/r06_stream_b9_equal_waves.txt): with every wave waiting at the same barrier, then issuing its DMAs, reading its operands, splitting
// its weights and multiplying - in that order, all sixteen in the same phase - the launch took the SUM of those phases (matrix pipe 34 % busy,
// 36 % of the issued instructions scalar bookkeeping), and each wave re-read all plane operands for ONE tile (168 KB of LDS reads per chunk).
// Here an MFMA wave holds HALF the workgroup's tiles x all columns of one k-block: the plane operands are read once per three tiles, and the
// split of the NEXT chunk's weight fragments (5.5 vector instructions per weight: and / subtract / byte-permute) is issued in the shadow of
// this chunk's MFMAs - a 16-clock MFMA leaves three issue slots, 108 MFMAs per chunk (three tiles x four column tiles x nine) hide 132 + 36
// vector instructions and 18 LDS reads.
//   * X arrives as three bf16 planes (written by the kernel that produces the rows: k_rmsnorm_rows_s3, the attention kernels, the SiLU
//     epilogue, k_stream_reduce_norm) - never converted on the matmul side;
//   * MFMA wave w = 2 kb + part: k-block kb (32 columns) of every chunk, tiles [0, ceil(MAXT / 2)) or [ceil(MAXT / 2), MAXT), all column
//     tiles (odd tile counts next to an even column-tile count: every tile, half the column tiles - CSP); the MFMA is issued transposed (A = activations, B = weights: a lane's four results belong to ONE weight row);
//   * the two k-block partial tiles of a tile meet in LDS and are added in wave order (stream_epilogue: bit-reproducible), then the launch's
//     epilogue: + residual | silu(w1 h) * (w3 h) on (w1, w3) tile pairs | RoPE + cache append - and optionally the result split into planes
//     for the next launch.
//   image per chunk: weights [MAXT * 16 rows][64 floats], dense (the DMA writes lane-linearly), 16-byte granule g of row r at position
//     g ^ (r & 15) (a ds_read_b128 lane group - eight rows at slot s, eight at slot s ^ 1 - touches sixteen different positions); planes
//     [3][NCT * 16 rows][64 bf16], 128-byte rows, granule g of row r at g ^ ((r >> 1) & 7) (two rows per bank line).
// Values: every product is exact for finite inputs whose three parts are normal bf16 numbers (inputs above ~2^-110: kernels_gemm_b9.h).
#pragma once
#include "kernels_stream_q8b.h"

namespace lh {

// eight fp32 weights -> the three bf16 pieces of each, packed in k order (element j in half j & 1 of dword j >> 1: the planes' memory order)
__device__ __forceinline__ void split3x8(const f4 a, const f4 b, u4* hi, u4* mid, u4* lo) {
    const float w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t r1[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const uint32_t wb = __builtin_bit_cast(uint32_t, w[j]);
        const float r = __fsub_rn(w[j], __builtin_bit_cast(float, wb & 0xffff0000u));          // exact: the low 16 bits of the significand
        r1[j] = __builtin_bit_cast(uint32_t, r);
        r2[j] = __builtin_bit_cast(uint32_t, __fsub_rn(r, __builtin_bit_cast(float, r1[j] & 0xffff0000u)));   // exact: <= 8 significant bits
    }
    // v_perm_b32 packs two high halves (the truncation of hi and mid costs no instruction of its own)
#define B9_PK(x1, x0) __builtin_amdgcn_perm((x1), (x0), 0x07060302u)
    *hi = u4{B9_PK(__builtin_bit_cast(uint32_t, w[1]), __builtin_bit_cast(uint32_t, w[0])), B9_PK(__builtin_bit_cast(uint32_t, w[3]), __builtin_bit_cast(uint32_t, w[2])),
             B9_PK(__builtin_bit_cast(uint32_t, w[5]), __builtin_bit_cast(uint32_t, w[4])), B9_PK(__builtin_bit_cast(uint32_t, w[7]), __builtin_bit_cast(uint32_t, w[6]))};
    *mid = u4{B9_PK(r1[1], r1[0]), B9_PK(r1[3], r1[2]), B9_PK(r1[5], r1[4]), B9_PK(r1[7], r1[6])};
    *lo = u4{B9_PK(r2[1], r2[0]), B9_PK(r2[3], r2[2]), B9_PK(r2[5], r2[4]), B9_PK(r2[7], r2[6])};
#undef B9_PK
}

// MAXT: 16-row weight tiles per workgroup; NCT: 16-token column tiles (<= 4); NIMG: images in the ring;
// NPROD: 9 (exact) - probe builds: 8 drops xl * wl, 6 also xm * wl and xl * wm (tools/b9s_probe: what each costs against an f64 product)
constexpr int B9S_TH = 512, B9S_KC = 64;
__host__ __device__ constexpr size_t stream_b9_image_bytes(int maxt, int nct) { return (size_t)maxt * 16 * B9S_KC * 4 + (size_t)3 * nct * 16 * B9S_KC * 2; }
__host__ __device__ constexpr int stream_b9_nimg(int maxt, int nct, int cap) {
    const int n = (int)(160 * 1024 / stream_b9_image_bytes(maxt, nct));
    return n < cap ? n : cap;
}
#ifndef B9S_ABLATE
#define B9S_ABLATE 0   // tools/b9s_probe timing-only builds: 2 no weight split | 4 one product | 8 no DMAs behind the prologue
#endif
// CSP: the two MFMA waves of a k-block halve the COLUMN tiles (every tile each) instead of the tiles - for odd tile counts (wq|wk|wv of 7B: three
// tiles per workgroup, which halved by tiles leave two SIMDs two thirds of the work); both waves then split the same weight fragments.
__host__ __device__ constexpr bool stream_b9_csp(int maxt, int nct) { return (maxt & 1) && nct % 2 == 0; }
template <int MAXT, int NCT, int NIMG, int NPROD = 9, bool CSP = stream_b9_csp(MAXT, NCT)>
__global__ __launch_bounds__(B9S_TH) void k_stream_b9(const StreamArgs a) {
    LH_TOUCH_ARGS(a.w[0], a.r[2], a.epi, a.gamma, a.ys_plane, a.ldys);   // the argument block's lines behind one wait (kernels_common.h)
    constexpr int KC = B9S_KC;
    static_assert(NIMG >= 2 && NIMG <= 5, "ring");
    static_assert(NCT >= 1 && NCT <= 4, "column tiles");
    static_assert(NPROD == 9 || NPROD == 8 || NPROD == 6, "products");
    constexpr int XR = NCT * 16;                // staged activation rows per plane
    constexpr int GRW = 16, RPW = 4;            // 16-byte granules per weight row; weight rows per DMA instruction
    constexpr int GRX = 8, RPX = 8;             // granules per plane row; plane rows per DMA instruction
    constexpr int NWI = MAXT * 16 / RPW, NXP = XR / RPX, NXI = 3 * NXP, NI = NWI + NXI;
    constexpr int NIW = (NI + 3) / 4;           // DMA instructions per loader wave and chunk
    constexpr int WAITN = NIW * (NIMG - 2) < 64 ? NIW * (NIMG - 2) : 63;
    constexpr uint32_t W_BYTES = MAXT * 16 * KC * 4, XP_BYTES = XR * KC * 2, IMG_BYTES = W_BYTES + 3 * XP_BYTES;
    static_assert(!CSP || NCT % 2 == 0, "column halves");
    constexpr int T0 = CSP ? MAXT : (MAXT + 1) / 2, TPW = T0;   // tiles of part 0 = tile slots of an MFMA wave
    constexpr int NCW = CSP ? NCT / 2 : NCT;    // column tiles of an MFMA wave
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tiles_per_mat = a.M >> 4, T = tiles_per_mat * a.groups;
    const bool pairs = a.epi == ST_EPI_SILU_MUL;   // virtual tile v = (tile v >> 1 of matrix v & 1), dealt in PAIRS (k_stream_mm2)
    const uint32_t units = pairs ? tiles_per_mat : T, um = pairs ? 2u : 1u;
    const uint32_t S = a.ksplit > 1 ? a.ksplit : 1u, bg = (uint32_t)blockIdx.x / S, ks = (uint32_t)blockIdx.x - bg * S, ng = (uint32_t)gridDim.x / S;
    if (bg >= ng) return;
    const uint32_t t0 = um * (bg * units / ng), t1 = um * ((bg + 1) * units / ng);
    if (t1 <= t0) return;
    const uint32_t nt = t1 - t0;                // <= MAXT (host)
    const uint32_t nch_all = a.K / KC, ch0 = ks * nch_all / S;
    const uint32_t nch = (ks + 1) * nch_all / S - ch0;
    const uint32_t kbase = ch0 * KC;
    if (nch == 0) return;
    const uint32_t r16 = (uint32_t)lane & 15, slot = (uint32_t)lane >> 4;
    auto xswz = [](uint32_t r) -> uint32_t { return (r >> 1) & 7u; };
#ifdef Q8B_TRACE   // tools/b9s_probe: shader clocks and 100 MHz ticks of one workgroup's life (what the chip clocks at under this kernel)
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    f4m acc[TPW][NCW];
    if (wave < 4) {
        // ---- loader waves: piece q = 4 j + wave of a chunk (the tiles' weight rows, then the three planes); a piece's lanes sit in ONE 16-row
        // tile / one plane, so its first row is a uniform resource base and the lane keeps (row) * pitch + its swizzled granule
        uint32_t voff[NIW], doff[NIW], sstep[NIW];
        const char* base[NIW];
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            uint32_t q = (uint32_t)j * 4 + (uint32_t)wave;
            q = q < (uint32_t)NI ? q : (uint32_t)NI - 1;                   // surplus slots repeat the last piece (same bytes to the same place)
            if (q < (uint32_t)NWI) {
                const uint32_t rr = q * RPW + (uint32_t)lane / GRW, gd = (uint32_t)lane % GRW, gs = gd ^ (rr & 15u);
                uint32_t ts = (q * RPW) >> 4;                              // tile slot of the instruction: uniform
                ts = ts < nt ? ts : nt - 1;                                // slots beyond the block: a valid tile again, its sums are never stored
                const uint32_t v = t0 + ts;
                uint32_t g, tile;
                if (pairs) { g = v & 1u; tile = v >> 1; }
                else { g = (v >= tiles_per_mat ? 1u : 0u) + (v >= 2 * tiles_per_mat ? 1u : 0u); tile = v - g * tiles_per_mat; }   // (<= 3 matrices: no division)
                base[j] = (const char*)((g == 0 ? a.w[0] : (g == 1 ? a.w[1] : a.w[2])) + (size_t)tile * 16 * a.K + kbase);
                voff[j] = ((rr & 15u) * a.K + gs * 4u) * 4u;
                sstep[j] = KC * 4;
                doff[j] = q * 1024u;
            } else {
                const uint32_t xq = q - NWI, p = xq / NXP, xi = xq - p * NXP;
                const uint32_t rr = xi * RPX + (uint32_t)lane / GRX, gd = (uint32_t)lane % GRX, gs = gd ^ xswz(rr);
                const uint32_t c = rr < a.n ? rr : a.n - 1;                // rows past the batch: the last row again (never stored)
                base[j] = (const char*)(a.xs + (size_t)p * a.xs_plane + kbase);
                voff[j] = (c * a.ldxs + gs * 8u) * 2u;
                sstep[j] = KC * 2;
                doff[j] = W_BYTES + p * XP_BYTES + xi * 1024u;
            }
        }
        auto issue = [&](uint32_t ch) {
            const uint32_t cc = ch < nch ? ch : nch - 1;                   // past the end: a harmless reload into a free image (uniform counts)
            char* im = smem_raw + (size_t)(ch % NIMG) * IMG_BYTES;
            // (resource and destination as named locals: kernels_stream.h, k_stream_dma)
#pragma unroll
            for (int j = 0; j < NIW; ++j) {     // 1 KB per instruction
                const __amdgpu_buffer_rsrc_t rs = stream_rsrc(base[j]);
                __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(im + doff[j]);
                const int so = (int)(cc * sstep[j]);
                if (doff[j] < W_BYTES) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (int)voff[j], so, 0, 2);   // weights: read once (nt)
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (int)voff[j], so, 0, 0);                     // planes: every workgroup reads them, out of L2
            }
        };
#pragma unroll
        for (int c = 0; c < NIMG - 1; ++c) issue((uint32_t)c);
        for (uint32_t ch = 0; ch < nch; ++ch) {
            wait_vm<WAITN>();                   // chunk ch of this wave has landed; the NIMG - 2 younger ones may still be in flight
            __builtin_amdgcn_s_barrier();       // barrier ch: every part of chunk ch is in its image, and the MFMA waves have left chunk ch - 1's
            if (!(B9S_ABLATE & 8) || ch + NIMG - 1 < (uint32_t)NIMG) issue(ch + NIMG - 1);   // ... whose image takes chunk ch + NIMG - 1
        }
        __builtin_amdgcn_s_barrier();           // the MFMA waves' extra period (the last chunk's MFMAs)
        wait_vm<0>();                           // the clamped tail requests
    } else {
        // ---- MFMA waves
        const uint32_t cw = (uint32_t)(wave - 4), kb = cw >> 1, part = cw & 1u, tb = CSP ? 0u : part * T0, cb = CSP ? part * NCW : 0u;
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int c = 0; c < NCW; ++c) acc[j][c] = f4m{0.f, 0.f, 0.f, 0.f};
        uint32_t woff[TPW], xoff[NCW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            uint32_t t = tb + (uint32_t)j;
            t = t < (uint32_t)MAXT ? t : (uint32_t)MAXT - 1;               // (part 1 of an odd tile count: its last slot repeats a tile, never stored)
            woff[j] = (t * 16 + r16) * (KC * 4) + (((kb * 8 + slot * 2) ^ r16) * 16);   // the fragment's first granule; the second at position ^ 1
        }
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
            const uint32_t row = (cb + (uint32_t)c) * 16 + r16;
            xoff[c] = W_BYTES + (row * GRX + ((kb * 4 + slot) ^ xswz(row))) * 16;
        }
        auto image = [&](uint32_t ch) { return (const char*)(smem_raw + (size_t)(ch % NIMG) * IMG_BYTES); };
        u4 wsp[TPW][3], spare[3];               // this chunk's weight fragments as three bf16 pieces each; the fragment being prepared
        u4 xa[NCW][3], xb[NCW][3];              // two operand sets of the planes, swapped by name
        float raw[TPW][8];                      // the next chunk's fragments as they come out of LDS
        uint32_t s1[8], s2[8];                  // a fragment's residuals on their way into `spare`
        // One period, scheduled BY HAND (the compiler's sched_group_barrier solver gives the interleave up as soon as LDS reads stand among the
        // groups - r06_stream_b9_probe.txt - and with all 18 reads of a wave in one burst behind the barrier the matrix pipe waited a fifth of the
        // period): the operands of chunk ch are in registers (wsp, xc); after every MFMA, fenced by sched_barrier(0), ONE slot of filler work for
        // chunk ch + 1 - whose image was completed a barrier ago:
        //   slots 0 .. NDS - 1 of the period      the 2 TPW + 3 NCW operand reads (fragments first)
        //   tile j's slots from S0 on              fragment j: eight 4-instruction splits (and, subtract, and, subtract), then twelve byte-permutes
        //                                          into `spare` (tile j's MFMAs still read wsp[j])
        //   the first slots of tile j + 1          `spare` -> wsp[j] (behind tile j's last MFMA)
        // (No branch on "is there a next chunk": behind the last one the reads take whatever the ring's next image holds - the loader's clamped
        // tail requests - and the fragments made of it are never multiplied.)
        constexpr int NM = 9 * NCW, NMF = NM * TPW, NDS = 2 * TPW + 3 * NCW;
        constexpr int P0 = (B9S_ABLATE & 4) ? 8 : (NPROD == 9 ? 0 : (NPROD == 8 ? 1 : 3));
        auto ds_slot = [&](int e, const char* im, u4 (&xn)[NCW][3]) {
            if (e < 2 * TPW) {
                const f4 v = *(const f4*)(im + (woff[e >> 1] ^ ((e & 1) ? 16u : 0u)));
                raw[e >> 1][4 * (e & 1)] = v.x; raw[e >> 1][4 * (e & 1) + 1] = v.y; raw[e >> 1][4 * (e & 1) + 2] = v.z; raw[e >> 1][4 * (e & 1) + 3] = v.w;
            } else {
                const int c = (e - 2 * TPW) / 3, pl = (e - 2 * TPW) % 3;
                xn[c][pl] = *(const u4*)(im + xoff[c] + (size_t)pl * XP_BYTES);
            }
        };
        auto split_f = [&](int j, int f) {      // 4 vector instructions
            const uint32_t hb = __builtin_bit_cast(uint32_t, raw[j][f]) & 0xffff0000u;
            const float r1 = __fsub_rn(raw[j][f], __builtin_bit_cast(float, hb));
            const uint32_t mb = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
            s1[f] = __builtin_bit_cast(uint32_t, r1); s2[f] = __builtin_bit_cast(uint32_t, __fsub_rn(r1, __builtin_bit_cast(float, mb)));
            asm volatile("" : "+v"(s1[f]), "+v"(s2[f]));   // (an empty asm that "uses" the results: without it the optimiser sinks the arithmetic out of its slot, below the fences)
        };
        auto pack1 = [&](int j, int k) {        // one byte-permute: dword k & 3 of plane k >> 2
            const int pl = k >> 2, d = k & 3;
            if constexpr ((B9S_ABLATE & 2) != 0) { spare[pl][d] = __builtin_bit_cast(uint32_t, raw[j][2 * d]); return; }
            const uint32_t lo = pl == 0 ? __builtin_bit_cast(uint32_t, raw[j][2 * d]) : (pl == 1 ? s1[2 * d] : s2[2 * d]);
            const uint32_t hi = pl == 0 ? __builtin_bit_cast(uint32_t, raw[j][2 * d + 1]) : (pl == 1 ? s1[2 * d + 1] : s2[2 * d + 1]);
            spare[pl][d] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);
            asm volatile("" : "+v"(spare[pl]));
        };
        auto period = [&](uint32_t ch, const u4 (&xc)[NCW][3], u4 (&xn)[NCW][3]) {
            const char* im = image(ch + 1);
            constexpr int PX[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0}, PW[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0};
            constexpr int NMT = (9 - P0) * NCW;                        // MFMAs per tile
            constexpr int S0 = NMT >= 30 ? 10 : (NMT >= 20 ? 6 : 2);   // first split slot of a tile (the fragment's way from LDS; the moves of the tile before)
            constexpr int NPC = 20;                                    // pieces per fragment: 8 splits + 12 permutes
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
#pragma unroll
                for (int m = 0; m < NMT; ++m) {
                    const int q = P0 + m / NCW, c = m % NCW, e = j * NMT + m;
                    acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xc[c][PX[q]]), __builtin_bit_cast(bf16x8, wsp[j][PW[q]]), acc[j][c], 0, 0, 0);
                    if (e < NDS) ds_slot(e, im, xn);
                    else if (NDS > NMT * TPW && m == NMT - 1 && j == TPW - 1) {   // (fewer MFMAs than reads: the rest behind the last one)
#pragma unroll
                        for (int r = NMT * TPW; r < NDS; ++r) ds_slot(r, im, xn);
                    }
                    if (j > 0 && m < 6) {                                  // tile j - 1's fragment into place, two dwords per slot
                        const int k0 = 2 * m;
                        wsp[j - 1][k0 >> 2][k0 & 3] = spare[k0 >> 2][k0 & 3]; wsp[j - 1][(k0 + 1) >> 2][(k0 + 1) & 3] = spare[(k0 + 1) >> 2][(k0 + 1) & 3];
                        asm volatile("" : "+v"(wsp[j - 1][k0 >> 2]));
                    }
                    if (m >= S0) {
                        const int sl = m - S0, nsl = NMT - S0;
#pragma unroll
                        for (int pc = sl * NPC / nsl; pc < (sl + 1) * NPC / nsl; ++pc) {
                            if (pc < 8) { if constexpr ((B9S_ABLATE & 2) == 0) split_f(j, pc); }
                            else pack1(j, pc - 8);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wsp[TPW - 1][pl] = spare[pl];   // the last tile's fragment (behind its last MFMA)
            __builtin_amdgcn_sched_barrier(0);
        };
        barrier_lds_only();                     // barrier 0: chunk 0 is in image 0
#pragma unroll
        for (int e = 0; e < NDS; ++e) ds_slot(e, smem_raw, xa);
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
#pragma unroll
            for (int f = 0; f < 8; ++f) if constexpr ((B9S_ABLATE & 2) == 0) split_f(j, f);
#pragma unroll
            for (int k = 0; k < 12; ++k) pack1(j, k);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wsp[j][pl] = spare[pl];
        }
        __builtin_amdgcn_sched_barrier(0);
        for (uint32_t ch = 0; ch < nch; ch += 2) {   // two chunks per trip: the operand sets swap by name
            barrier_lds_only();                 // barrier ch + 1
            period(ch, xa, xb);
            if (ch + 1 < nch) {
                barrier_lds_only();             // barrier ch + 2
                period(ch + 1, xb, xa);
            }
        }
        if (!(nch & 1)) {}                      // (an even chunk count ends on barrier nch, an odd one too: the loader's extra barrier pairs with the last one taken above)
    }
    __syncthreads();   // the images are dead (the loader waves have drained their DMAs: a pending LDS-DMA would land in `part`)
#ifdef Q8B_TRACE
    if (a.trace && blockIdx.x == gridDim.x / 2 && tid == 256) { a.trace[0] = __builtin_amdgcn_s_memtime() - clk_c0; a.trace[1] = __builtin_amdgcn_s_memrealtime() - clk_r0; }
#endif
    if constexpr (CSP) stream_epilogue<MAXT, NCT, 2, true>(a, smem_raw, (uint32_t)((size_t)NIMG * IMG_BYTES / 4), nullptr, t0, nt, ks, tiles_per_mat, [&](int t, int c) { return acc[t][c]; });
    else stream_epilogue<MAXT, NCT, 1, true, 0, 2>(a, smem_raw, (uint32_t)((size_t)NIMG * IMG_BYTES / 4), nullptr, t0, nt, ks, tiles_per_mat,
                                                   [&](int t, int c) { return acc[t >= T0 ? t - T0 : t][c]; });
}

}  // namespace lh
