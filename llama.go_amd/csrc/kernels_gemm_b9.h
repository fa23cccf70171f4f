// csrc/kernels_gemm_b9.h — the prefill GEMM on the bf16 matrix pipe with EXACT fp32 products (round 5; BASELINE config 3).
//     Y_g[n][m] (+ R_g[n][m]) = sum_k X[n][k] * W_g[m][k]        (ComputeForwardMulMatFP32, pkg/ml/ml.go:1976-2098; builder ml.go:295-318)
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate - 64 clocks per SIMD for 32 x 32 x 2 - and k_gemm_glds holds 0.80 of that peak
// (profiles/r04_mfma_clock.txt).  v_mfma_f32_32x32x16_bf16 does 32 x 32 x 16 in 32 clocks: 16x the rate.  An fp32 number is exactly the sum
// of three bf16 (8 + 8 + 8 significand bits: split3, kernels_stream.h), so
//     x * w = (xh + xm + xl) * (wh + wm + wl) = nine products of 8-bit significands, each EXACT in fp32,
// i.e. nine bf16 MFMAs (288 clocks) do the work of eight fp32 MFMAs (512 clocks) with no narrow-precision input anywhere: what differs from
// the fp32 instruction is only the order in which exact products meet in the fp32 accumulator (small terms first here), as any tiling changes
// it.  SURVEY App. C forbids LOSSY narrow inputs; this is the lossless split the block-int8 kernel uses (kernels_stream_q8b.h), on both sides.
//
// Structure: k_gemm_glds's (persistent one workgroup per CU, XCD-aware tile order, ring of LDS stages fed by global -> LDS DMA, one
// workgroup barrier per 32-column slab, the same C layout and therefore the same epilogues gemm_store / gemm_store_fused).  Differences:
//   * X arrives already split (three bf16 planes, k_split3_rows / k_rmsnorm_rows_s3): its slab is [3][BN rows][32 bf16], 16-byte granule g of
//     row r at g ^ ((r >> 2) & 3) (a ds_read_b128 lane group = 16 rows x one k-group);
//   * W stays fp32 in HBM and in LDS ([BM rows][32 floats], k_gemm_glds's swizzle) and is split on the operand-read side: per k-step of 16 a
//     lane reads 8 floats (two ds_read_b128) and makes 3 x 8 bf16 out of them with 32 and / subtract + 12 byte-permutes - 44 vector
//     instructions per weight fragment against 9 TN MFMAs of 32 clocks.  Vector instructions are NOT free next to the MFMAs (each costs
//     the matrix pipe about its own 4 clocks, measured by leaving the split out), so the product shape gives a wave ALL BN = 128 rows
//     (TN = 4): no two waves split the same weights and a fragment's 44 instructions stand against 36 MFMAs.
// Per k-step and wave: TN x 3 + TM x 2 LDS reads, TM x 44 vector instructions, 9 TN TM MFMAs.
// Product instantiation <1, 8, 4, 1, 2>: 128 x 256 tiles, eight waves of 128 x 32, two stages of 56 KB.
//
// What bounds it (profiles/r05_gemm_b9_probe.txt): POWER.  In clocks the kernel is where it should be - 13B w1|w3 at 1024 rows spends 3.29 M
// shader clocks per workgroup of which 2.95 M are MFMA issue (0.90) - but with all 256 CUs on the bf16 pipe the chip holds 1.44-1.74 GHz
// (2.15-2.18 GHz with 160 CUs busy; 2.39 GHz under k_gemm_glds' fp32 MFMAs, r04_mfma_clock.txt).  Net: 1.04-1.21x k_gemm_glds' best shape
// on the 13B matrices, 206 -> 191-197 ms for the 40-layer 1024-token prompt.  Scheduling is by hand (see the loop): the compiler's
// sched_group_barrier solver gave up the interleave when one group could not be filled, and through the LDS-DMA builtin it drains the ring
// (vmcnt(0)) before every operand read, so the DMA is inline asm.
// Values: every product is exact for finite inputs whose three parts are normal bf16 numbers; parts below 2^-126 (inputs below ~2^-110)
// are at the mercy of the matrix pipe's denormal handling, far below anything a model's activations or weights hold.
#pragma once
#include "kernels_gemm.h"
#include "kernels_stream.h"

namespace lh {

typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));

constexpr int B9_GST = 3;
__host__ __device__ constexpr size_t gemm_b9_stage_bytes(int bn, int bm) { return (size_t)3 * bn * 64 + (size_t)bm * 128; }

// WN x WM waves (4 or 8: two waves per SIMD let one wave's weight split run under the other's MFMAs), each TN x TM tiles of 32 x 32
template <int WN, int WM, int TN, int TM, int NST = B9_GST>
__global__ __launch_bounds__(WN * WM * 64) void k_gemm_b9(const GemmArgs a) {
    constexpr int NWV = WN * WM;
    static_assert(NWV == 4 || NWV == 8, "waves");
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    constexpr int XP_BYTES = BN * 64, W_OFF = 3 * XP_BYTES, STAGE = W_OFF + BM * 128;
    constexpr int XPIECES = 3 * BN / 16, WPIECES = BM / 8, PIECES = XPIECES + WPIECES, PPW = (PIECES + NWV - 1) / NWV;
    extern __shared__ __attribute__((aligned(16))) char smem_b9[];  // [NST][STAGE]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WM, wm = wave % WM;
    const uint32_t tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    // split-K as in k_gemm_glds (launches whose tiles fill the CUs badly: 160 tiles of the 13B wo / w2, 128 of the 7B ones): work item = (tile,
    // slab range ks), partial products to part[group][ks][N][M], k_splitk_reduce adds them in ks order (+ residual); ranges differ by <= 1 slab
    const uint32_t splits = a.splits ? a.splits : 1;
    const uint32_t per_group = tiles_n * tiles_m, total = per_group * a.groups * splits;
    const uint32_t ldw = a.ldw ? a.ldw : a.K;
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t nkf = a.K / GBK;
    // persistent workgroups, one per CU; the workgroups of one XCD run consecutive tiles, n fastest (k_gemm_glds's order: a weight panel meets in one L2)
    const uint32_t G = gridDim.x;
    const uint32_t v0 = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
#ifdef B9_TRACE
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (uint32_t wv = v0; wv < total; wv += G) {
        const uint32_t ks = wv % splits, wi = wv / splits;      // K range fastest: the pieces of one tile run side by side
        const uint32_t g = wi / per_group, t = wi % per_group;
        const uint32_t tm = t / tiles_n, tn = t % tiles_n;
        const uint32_t n0 = tn * BN, m0 = tm * BM;
        const uint32_t sl0 = (uint32_t)(((uint64_t)ks * nkf) / splits), nk = (uint32_t)(((uint64_t)(ks + 1) * nkf) / splits) - sl0;   // first slab, slabs
        __builtin_amdgcn_s_barrier();  // every wave is done reading the previous tile's stages
        // piece p of a slab is fetched by wave p % NWV: this lane's source pointer and LDS offset for each of its pieces
        const char* src[PPW];
        uint32_t dst[PPW], kstep[PPW];
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) {
            uint32_t p = (uint32_t)wave + (uint32_t)NWV * pp;
            p = p < (uint32_t)PIECES ? p : (uint32_t)PIECES - 1;              // surplus slots repeat the last piece (same bytes to the same place)
            if (p < (uint32_t)XPIECES) {
                const uint32_t pl = p / (BN / 16), row = (p % (BN / 16)) * 16 + (uint32_t)(lane >> 2);   // plane, tile row
                const uint32_t gs = (uint32_t)(lane & 3) ^ ((row >> 2) & 3u);                             // source granule stored at position lane & 3
                const uint32_t n = n0 + row;
                src[pp] = (const char*)(a.xs + (size_t)pl * a.xs_plane + (size_t)(n < a.N ? n : a.N - 1) * a.ldxs) + gs * 16;
                dst[pp] = pl * XP_BYTES + (p % (BN / 16)) * 1024;
                kstep[pp] = GBK * 2;
            } else {
                const uint32_t q = p - XPIECES, row = q * 8 + (uint32_t)(lane >> 3);
                const uint32_t gs = (uint32_t)(lane & 7) ^ ((row >> 1) & 7u);
                const uint32_t m = m0 + row, mm = m < a.M ? m : a.M - 1;
                if (a.epi == GEMM_EPI_SILU_MUL)   // virtual row mm = row mm >> 1 of w1 (even) / w3 (odd); base + distance, never a pointer select
                    src[pp] = (const char*)((const float*)((uint64_t)a.w[0] + ((mm & 1u) ? (uint64_t)a.w[1] - (uint64_t)a.w[0] : 0)) + (size_t)(mm >> 1) * ldw) + gs * 16;
                else
                    src[pp] = (const char*)(a.w[g] + (size_t)mm * ldw) + gs * 16;
                dst[pp] = W_OFF + q * 1024;
                kstep[pp] = GBK * 4;
            }
        }
        // (the DMA is inline asm on purpose - kernels_stream_q8b.h: through the builtin the compiler sees a store to LDS and drains every DMA in
        // flight, s_waitcnt vmcnt(0), in front of this wave's next operand read; the waits for these are the counted ones of the loop)
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_b9;
        uint32_t dstS[PPW];                                                   // wave-uniform: kept in scalar registers
#pragma unroll
        for (int pp = 0; pp < PPW; ++pp) dstS[pp] = (uint32_t)__builtin_amdgcn_readfirstlane((int)dst[pp]);
        auto issue1 = [&](uint32_t slab, int pp) {
            const uint32_t ks = sl0 + (slab < nk ? slab : nk - 1);            // past the end: the last slab again (keeps the counts uniform; harmless)
            const uint32_t st = lds0 + (slab % NST) * STAGE;
            const char* sp = src[pp] + (size_t)ks * kstep[pp];
            const uint32_t m0v = st + dstS[pp];
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(sp) : "memory", "m0");
        };
        auto issue = [&](uint32_t slab) {
#pragma unroll
            for (int pp = 0; pp < PPW; ++pp) issue1(slab, pp);
        };
        f16v acc[TN][TM];
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const uint32_t xrow = (uint32_t)(wn * TN * 32 + li) * 64, xsw = ((uint32_t)li >> 2) & 3u;
        const uint32_t wrow = W_OFF + (uint32_t)(wm * TM * 32 + li) * 128, wsw = ((uint32_t)li >> 1) & 7u;
        // Software pipeline over the k-steps (two per slab), scheduled BY HAND: while step s multiplies out of registers, the operands of
        // step s + 1 are read and its weight fragments split, a few instructions behind each MFMA, every group fenced by sched_barrier(0)
        // (sched_group_barrier's solver gave the interleave up whenever one of its groups could not be filled - r05_gemm_b9_probe.txt).
        // Slab boundary, inside the SECOND step of slab kt after its first BQ MFMAs (the time slower waves have to catch up): every wave
        // has by then read all of slab kt into registers, so behind "my pieces of slab kt + 1 have landed" + s_barrier the next step's
        // operands are read from slab kt + 1 and slab kt's stage is refilled with slab kt + GST, one DMA behind each of the next MFMAs.
        u4 xa[2][3][TN], wq[2][3][TM];     // [operand set][plane][tile]
        float wr[TM][8];                   // the weight floats of the step being prepared
        uint32_t sb[3][TM][8];             // their hi / mid / lo parts before packing
        auto read_w = [&](uint32_t slab, int kk, int j, int half) {
            const char* st = smem_b9 + (size_t)(slab % NST) * STAGE;
            const f4 w = *(const f4*)(st + wrow + j * 32 * 128 + (((uint32_t)(4 * kk + 2 * lh + half)) ^ wsw) * 16);
            wr[j][4 * half] = w.x; wr[j][4 * half + 1] = w.y; wr[j][4 * half + 2] = w.z; wr[j][4 * half + 3] = w.w;
        };
        auto read_x = [&](int set, uint32_t slab, int kk, int pl, int i) {
            const char* st = smem_b9 + (size_t)(slab % NST) * STAGE;
            xa[set][pl][i] = *(const u4*)(st + pl * XP_BYTES + xrow + i * 32 * 64 + (((uint32_t)(2 * kk + lh)) ^ xsw) * 16);
        };
        auto split_f = [&](int j, int f) {              // 4 vector instructions
            const uint32_t hb = __builtin_bit_cast(uint32_t, wr[j][f]) & 0xffff0000u;
            const float r1 = __fsub_rn(wr[j][f], __builtin_bit_cast(float, hb));
            const uint32_t mb = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
            const float r2 = __fsub_rn(r1, __builtin_bit_cast(float, mb));
            sb[0][j][f] = __builtin_bit_cast(uint32_t, wr[j][f]); sb[1][j][f] = __builtin_bit_cast(uint32_t, r1); sb[2][j][f] = __builtin_bit_cast(uint32_t, r2);
        };
        auto pack2 = [&](int set, int j, int c) {       // 2 byte-permutes: dwords 2 c, 2 c + 1 of plane c / 2 ... (6 calls per fragment)
            const int pl = c >> 1, d = (c & 1) * 2;
            uint32_t* o = (uint32_t*)&wq[set][pl][j];
            o[d] = __builtin_amdgcn_perm(sb[pl][j][2 * d + 1], sb[pl][j][2 * d], 0x07060302u);
            o[d + 1] = __builtin_amdgcn_perm(sb[pl][j][2 * d + 3], sb[pl][j][2 * d + 2], 0x07060302u);
        };
        // (round 6) EIGHT of the nine products: xl * wl - at most 2^-32 of its product, far below the fp32 accumulator's own rounding - is dropped.  The kernel is
        // power-bound (1.44-1.74 GHz with every CU on the bf16 pipe), so an MFMA saved is time saved; against an f64 product the error does not move
        // (profiles/r06_gemm_b9_products.txt; the same measurement on the stream kernel: profiles/r06_stream_b9_probe.txt).  -DB9_PRODUCTS=9 restores it.
#ifndef B9_PRODUCTS
#define B9_PRODUCTS 8
#endif
        constexpr int NPR = B9_PRODUCTS;
        static_assert(NPR == 8 || NPR == 9, "products");
        constexpr int NMF = NPR * TN * TM, NDS = 3 * TN + 2 * TM, NPC = 14 * TM;   // MFMAs, LDS reads, vector pieces (8 splits + 6 packs per fragment)
        constexpr int BQ = NMF / 4;
        // nine exact partial products per (x tile, w tile), the small ones first; consecutive MFMAs go to different accumulators
        auto mfma_q = [&](int set, int q) {
            constexpr int PX[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0}, PW[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0};
            const int r = q / (TN * TM) + (9 - NPR), i = (q % (TN * TM)) / TM, j = q % TM;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8g, xa[set][PX[r]][i]), __builtin_bit_cast(bf16x8g, wq[set][PW[r]][j]), acc[i][j], 0, 0, 0);
        };
        auto piece = [&](int ns, int p) {
            if (p < 8 * TM) split_f(p / 8, p % 8);
            else pack2(ns, (p - 8 * TM) / 6, (p - 8 * TM) % 6);
        };
        // one k-step: multiply operand set `set`, prepare set ^ 1 from (slab, kk); boundary = the slab switch described above happens inside
        auto step = [&](int set, uint32_t slab, int kk, bool boundary, uint32_t refill) {
            const int ns = set ^ 1;
            const int q0 = boundary ? BQ : 0;                       // first MFMA that has preparation work behind it
            const int vs = 2 * TM + 2;                              // vector work starts this many slots after the first weight read
            const int nslots = NMF - q0 - vs;
#pragma unroll
            for (int q = 0; q < NMF; ++q) {
                if (boundary && q == q0) {
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NST - 2)) : "memory");
                    barrier_lds_only();         // (+ lgkmcnt(0): this wave's own reads of slab kt are complete before anybody's DMA may overwrite it)
                    __builtin_amdgcn_sched_barrier(0);
                }
                mfma_q(set, q);
                const int e = q - q0;
                if (e >= 0) {
                    if (e < 2 * TM) read_w(slab, kk, e / 2, e & 1);
                    else if (e < NDS) read_x(ns, slab, kk, (e - 2 * TM) / TN, (e - 2 * TM) % TN);
                    if (boundary && e >= 1 && e - 1 < PPW) issue1(refill, e - 1);
                    if (e >= vs) {
                        const int s = e - vs;
#pragma unroll
                        for (int p = s * NPC / nslots; p < (s + 1) * NPC / nslots; ++p) piece(ns, p);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll
        for (int j = 0; j < NST; ++j) issue((uint32_t)j);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW * (NST - 1)) : "memory");      // slab 0 of mine
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < TM; ++j) { read_w(0, 0, j, 0); read_w(0, 0, j, 1); }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < TN; ++i) read_x(0, 0, 0, pl, i);
#pragma unroll
        for (int p = 0; p < NPC; ++p) piece(0, p);
        __builtin_amdgcn_sched_barrier(0);
        for (uint32_t kt = 0; kt < nk; ++kt) {
            step(0, kt, 1, false, 0);                                       // first step of slab kt; prepares its second step
            step(1, kt + 1 < nk ? kt + 1 : kt, 0, true, kt + NST);      // second step; prepares the first step of slab kt + 1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail DMA
        float* Y = splits > 1 ? a.part + ((size_t)(g * splits + ks) * a.N) * a.M : a.y[g];
        const float* R = splits > 1 ? nullptr : a.r[g];
        if (splits > 1) gemm_store<TN, TM>(a, acc, Y, R, n0 + wn * TN * 32, m0 + wm * TM * 32, li, lh, a.M);
        else if (a.epi != GEMM_EPI_STORE) gemm_store_fused<TN, TM>(a, acc, g, n0 + wn * TN * 32, m0 + wm * TM * 32, li, lh);
        else gemm_store<TN, TM>(a, acc, Y, R, n0 + wn * TN * 32, m0 + wm * TM * 32, li, lh, a.ldy);
    }
#ifdef B9_TRACE   // tools/gemm_b9_probe.hip: shader clocks and 100 MHz ticks one workgroup spent here
    if (a.clk && blockIdx.x == G / 2 && tid == 0) { a.clk[0] = __builtin_amdgcn_s_memtime() - clk_c0; a.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0; }
#endif
}

// ---- block-int8 weights, long prompts: k_gemm_q8b3 --------------------------------------------------------------------------------------
//     Y[n][m] (+ R[n][m]) = sum over quant blocks b of d[m][b] * sum_{k in b} q[m][k] * X[n][k]      (format ours; checker: dequantised fp32 weights)
// |q| <= 127 is exact in bf16 and X arrives as three exact planes, so a block's inner sum is THREE bf16 MFMAs with exact products, taken into a
// zeroed partial accumulator and folded into the running one with the block's scale (k_stream_q8b's arithmetic on a tile GEMM): 3 MFMAs of 32
// clocks per 32 x 32 x 16 block where the dequantising k_gemm_q8 spends 8 of 64.  One slab of the ring = 32 columns = one quant block.
// Shape: 128 x 256 tiles, eight waves of 128 x 32 (four 32 x 32 tiles each, B operand = the wave's 32 weight rows), NST stages of
//   X [3][128][32 bf16] (k_gemm_b9's layout) | W [256 rows][32 int8] (16-byte half h of row r at h ^ ((r >> 3) & 1)) = 32 KB,
// + three buffers [256][4] of scales: a DMA granule holds one row's scales of FOUR slabs, so slab s brings quarter s % 4 of the rows of scale
// group s / 4 + 1, and a group is complete one slab before its first use.
// Lane (li = lane & 31: weight row, lh = lane >> 5) holds, per slab, the 16 int8 of columns 16 lh .. 16 lh + 15 (ONE ds_read_b128); k-step
// st uses bytes 8 st .. 8 st + 7, so X granule 2 lh + st stands on the other side (any pairing of columns inside a block is the same sum).
// Schedule, per slab and wave: 24 MFMAs in twelve groups of two - tile pair (0, 1) through both k-steps x three planes (low plane first),
// then pair (2, 3) - each group fenced with sched_barrier(0).  Behind the MFMAs of a group: the two X reads of the group three ahead (a ring
// of four operand pairs; from group 9 on they come from the NEXT slab), four of the packed FMAs that fold the finished pair's partial sums
// (pair (2, 3)'s are folded under the next slab's first groups), and from group 8 on - behind "slab kt + 1 has landed" + s_barrier - the next
// slab's weight / scale reads, its int8 -> bf16 conversion and this wave's DMA pieces for slab kt + NST - 1 into the stage slab kt - 1 left.
constexpr int Q3_BN = 128, Q3_BM = 256, Q3_XP = Q3_BN * 64, Q3_WOFF = 3 * Q3_XP, Q3_STAGE = Q3_WOFF + Q3_BM * 32;
__host__ __device__ constexpr size_t gemm_q8b3_lds_bytes(int nst) { return (size_t)nst * Q3_STAGE + 3 * Q3_BM * 16; }
typedef float f2v __attribute__((ext_vector_type(2)));

template <int NST>
__global__ __launch_bounds__(512) void k_gemm_q8b3(const GemmArgs a) {
    static_assert(NST >= 4, "ring depth");
    constexpr int BN = Q3_BN, BM = Q3_BM, XP_BYTES = Q3_XP, W_OFF = Q3_WOFF, STAGE = Q3_STAGE, S_OFF = NST * Q3_STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem_q3[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: LDS destinations and the wave-0 branch)
    const int li = lane & 31, lh = lane >> 5;
    const uint32_t tiles_n = (a.N + BN - 1) / BN, tiles_m = (a.M + BM - 1) / BM;
    // split-K exactly as in k_gemm_glds (few tiles: prompts of a few hundred rows): work item = (tile, K range ks), partial products to
    // part[group][ks][N][M], k_splitk_reduce adds them in ks order (+ residual).  A range is a whole number of scale groups (4 slabs); the
    // ranges of one tile may differ by one group (344 slabs = 86 groups in eight ranges of 11 or 10).
    const uint32_t splits = a.splits ? a.splits : 1;
    const uint32_t per_group = tiles_n * tiles_m, total = per_group * a.groups * splits;
    const uint32_t nkf = a.K / GBK, ngr = nkf / 4;          // slabs and scale groups of the matrix
    const uint32_t G = gridDim.x;
    const uint32_t v0 = (G % 8 == 0) ? (blockIdx.x % 8) * (G / 8) + blockIdx.x / 8 : blockIdx.x;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_q3;
#ifdef B9_TRACE
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (uint32_t wv = v0; wv < total; wv += G) {
        const uint32_t ks = wv % splits, wi = wv / splits;      // K range fastest: the pieces of one tile run side by side
        const uint32_t g = wi / per_group, t = wi % per_group;
        const uint32_t tm = t / tiles_n, tn = t % tiles_n;
        const uint32_t gr0 = (uint32_t)(((uint64_t)ks * ngr) / splits), nsg = (uint32_t)(((uint64_t)(ks + 1) * ngr) / splits) - gr0;   // this item's scale groups
        const uint32_t n0 = tn * BN, m0 = tm * BM, ks0 = 4 * gr0, nk = 4 * nsg;                                                        // first slab, slabs
        __builtin_amdgcn_s_barrier();  // every wave is done reading the previous tile's stages
        // this wave's DMA pieces of a slab: X row block `wave` of the three planes, W rows 32 wave .. + 31, (wave 0) the slab's quarter of the scales
        const char* xsrc[3]; const char* wsrc;
        {
            const uint32_t row = (uint32_t)wave * 16 + (uint32_t)(lane >> 2), gs = (uint32_t)(lane & 3) ^ ((row >> 2) & 3u);
            const uint32_t n = n0 + row, nc = n < a.N ? n : a.N - 1;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) xsrc[pl] = (const char*)(a.xs + (size_t)pl * a.xs_plane + (size_t)nc * a.ldxs) + gs * 16 + (size_t)ks0 * 64;
            const uint32_t wr = (uint32_t)wave * 32 + (uint32_t)(lane >> 1), wh = (uint32_t)(lane & 1) ^ ((wr >> 3) & 1u);
            const uint32_t m = m0 + wr, mc = m < a.M ? m : a.M - 1;
            wsrc = (const char*)a.w[g] + (size_t)mc * a.K + wh * 16 + (size_t)ks0 * 32;
        }
        auto ssrc = [&](uint32_t qd) {     // (wave 0) this lane's row of scale quarter qd
            const uint32_t sm = m0 + qd * 64 + (uint32_t)lane, smc = sm < a.M ? sm : a.M - 1;
            return (const char*)(a.ws[g] + (size_t)smc * nkf + ks0);
        };
        auto dma = [&](uint32_t m0v, const char* sp) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(m0v), "v"(sp) : "memory", "m0");
        };
        // piece pc of slab `slab` (0..2: X planes, 3: W, 4: scales - wave 0 only)
        auto issue1 = [&](uint32_t slab, int pc) {
            const uint32_t ks = slab < nk ? slab : nk - 1;                  // past the end: the last slab again (uniform counts; harmless)
            const uint32_t st = lds0 + (slab % NST) * STAGE;
            if (pc < 3) dma(st + pc * XP_BYTES + (uint32_t)wave * 1024, xsrc[pc] + (size_t)ks * 64);
            else if (pc == 3) dma(st + W_OFF + (uint32_t)wave * 1024, wsrc + (size_t)ks * 32);
            else {
                const uint32_t sgp = slab / 4 + 1, sgc = sgp < nsg ? sgp : nsg - 1, qd = slab % 4;
                dma(lds0 + S_OFF + (sgp % 3) * (BM * 16) + qd * 1024, ssrc(qd) + (size_t)sgc * 16);
            }
        };
        f16v acc[4][1], ps[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][0][e] = 0.f; ps[i][e] = 0.f; }
        const uint32_t xrow = (uint32_t)li * 64, xsw = ((uint32_t)li >> 2) & 3u;
        const uint32_t wpos = W_OFF + (uint32_t)(wave * 32 + li) * 32 + (((uint32_t)lh ^ (((uint32_t)li >> 3) & 1u)) * 16);
        const uint32_t spos = S_OFF + (uint32_t)(wave * 32 + li) * 16;
        u4 xr[4][2];            // ring of X operand pairs
        u4 wq[2][2];            // [slab parity][k-step] converted weights
        u4 raw;                 // next slab's 16 int8
        float dsc[2] = {0.f, 0.f};     // [slab parity] block scale of this lane's weight row
        auto read_x = [&](uint32_t slab, int grp) {   // group 0..11 of a slab -> ring slot grp % 4
            const int pr = grp / 6, sp = grp % 6, st = sp / 3, pl = 2 - sp % 3;
            const char* sb = smem_q3 + (size_t)(slab % NST) * STAGE + pl * XP_BYTES + xrow + (((uint32_t)(2 * lh + st)) ^ xsw) * 16;
            xr[grp % 4][0] = *(const u4*)(sb + (2 * pr) * 32 * 64);
            xr[grp % 4][1] = *(const u4*)(sb + (2 * pr + 1) * 32 * 64);
        };
        auto read_w = [&](uint32_t slab, int par) {
            raw = *(const u4*)(smem_q3 + (size_t)(slab % NST) * STAGE + wpos);
            dsc[par] = *(const float*)(smem_q3 + spos + ((slab / 4) % 3) * (BM * 16) + (slab % 4) * 4);
        };
        auto convert = [&](int par, int part) {       // part 0 / 1: k-step's eight int8 -> bf16 (8 conversions + 4 byte-permutes), pinned where it stands
            if (part > 1) return;
            const int d0 = (int)(part ? raw.z : raw.x), d1 = (int)(part ? raw.w : raw.y);
            uint32_t c[8];
            c[0] = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d0)); c[1] = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d0 >> 8));
            c[2] = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d0 >> 16)); c[3] = __builtin_bit_cast(uint32_t, (float)(d0 >> 24));
            c[4] = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d1)); c[5] = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d1 >> 8));
            c[6] = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d1 >> 16)); c[7] = __builtin_bit_cast(uint32_t, (float)(d1 >> 24));
            u4 o = u4{__builtin_amdgcn_perm(c[1], c[0], 0x07060302u), __builtin_amdgcn_perm(c[3], c[2], 0x07060302u), __builtin_amdgcn_perm(c[5], c[4], 0x07060302u), __builtin_amdgcn_perm(c[7], c[6], 0x07060302u)};
            asm volatile("" : "+v"(o));
            wq[par][part] = o;
        };
        auto fold = [&](int tile, int quarter, float d) {   // acc += d * ps for four of the tile's sixteen values (two packed FMAs)
            const f2v d2 = {d, d};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = 4 * quarter + 2 * h;
                const f2v p2 = {ps[tile][e], ps[tile][e + 1]}, a2 = {acc[tile][0][e], acc[tile][0][e + 1]};
                f2v r2 = __builtin_elementwise_fma(p2, d2, a2);
                asm volatile("" : "+v"(r2));          // (keeps the FMA behind these MFMAs: left alone it sinks to the end of the loop body)
                acc[tile][0][e] = r2.x; acc[tile][0][e + 1] = r2.y;
            }
        };
        // one slab: multiply out of registers, prepare the next one
        auto slab_body = [&](uint32_t kt, int par, bool first) {
            const uint32_t nxt = kt + 1 < nk ? kt + 1 : kt;
#pragma unroll
            for (int grp = 0; grp < 12; ++grp) {
                const int pr = grp / 6, sp = grp % 6, st = sp / 3;
                if (grp == 8) {
                    if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (NST - 3)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NST - 3)) : "memory");
                    barrier_lds_only();
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tile = 2 * pr + h;
                    const f16v z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    ps[tile] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8g, xr[grp % 4][h]), __builtin_bit_cast(bf16x8g, wq[par][st]), sp == 0 ? z : ps[tile], 0, 0, 0);
                }
                if (grp + 3 < 12) read_x(kt, grp + 3); else read_x(nxt, grp + 3 - 12);
                if (grp < 4 && !first) { fold(2, grp, dsc[par ^ 1]); fold(3, grp, dsc[par ^ 1]); }          // the previous slab's pair (2, 3)
                if (grp >= 6 && grp < 10) { fold(0, grp - 6, dsc[par]); fold(1, grp - 6, dsc[par]); }
                if (grp == 8) read_w(nxt, par ^ 1);
                if (grp >= 9) convert(par ^ 1, grp - 9);
                if (grp >= 8 && grp < 11) issue1(kt + NST - 1, grp - 8);
                if (grp == 11) { issue1(kt + NST - 1, 3); if (wave == 0) issue1(kt + NST - 1, 4); }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // prologue: scale group 0, slabs 0 .. NST - 2
        if (wave == 0) {
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) dma(lds0 + S_OFF + qd * 1024, ssrc((uint32_t)qd));
        }
#pragma unroll
        for (int j = 0; j < NST - 1; ++j) {
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) issue1((uint32_t)j, pc);
            if (wave == 0) issue1((uint32_t)j, 4);
        }
        if (wave == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(5 * (NST - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NST - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        read_x(0, 0); read_x(0, 1); read_x(0, 2);
        read_w(0, 0);
        convert(0, 0); convert(0, 1);
        __builtin_amdgcn_sched_barrier(0);
        slab_body(0, 0, true);
        slab_body(1, 1, false);
        for (uint32_t kt = 2; kt < nk; kt += 2) {
            slab_body(kt, 0, false);
            slab_body(kt + 1, 1, false);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { fold(2, q, dsc[1]); fold(3, q, dsc[1]); }      // the last slab's pair (2, 3)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the redundant tail DMA
        if (splits > 1) gemm_store<4, 1>(a, acc, a.part + ((size_t)(g * splits + ks) * a.N) * a.M, nullptr, n0, m0 + (uint32_t)wave * 32, li, lh, a.M);
        else gemm_store<4, 1>(a, acc, a.y[g], a.r[g], n0, m0 + (uint32_t)wave * 32, li, lh, a.ldy);
    }
#ifdef B9_TRACE
    if (a.clk && blockIdx.x == G / 2 && tid == 0) { a.clk[0] = __builtin_amdgcn_s_memtime() - clk_c0; a.clk[1] = __builtin_amdgcn_s_memrealtime() - clk_r0; }
#endif
}

}  // namespace lh
