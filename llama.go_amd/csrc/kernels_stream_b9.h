// csrc/kernels_stream_b9.h — fp32 weights, 17..128 token rows per weight pass, on the bf16 matrix pipe with EXACT products (round 6).
//     Y_g[c][m] (+ R_g[c][m]) = sum_k X[c][k] * W_g[m][k]        (ComputeForwardMulMatFP32, pkg/ml/ml.go:1976-2098)
// The rows are a short prompt's tokens (server.Do feeds the prompt as ONE Eval, pkg/server/server.go:185-192) or the pods of a tick
// (server.go:84-106).
//
// Why: k_stream_dma (kernels_stream.h) multiplies on v_mfma_f32_16x16x4_f32, which issues at the fp32 VECTOR rate (32 clocks per SIMD for
// 16 x 16 x 4); from 33 rows on its launches are bound by the matrix pipe, not by the weight stream (64 rows: w1|w3 of 7B 110 us against a
// 53 us stream), and the chip drops to 2.09 GHz beside the HBM stream (profiles/r04_stream_eight_tiles_clock.txt).  An fp32 number is
// exactly the sum of three bf16 (8 + 8 + 8 significand bits: split3, kernels_stream.h), so
//     x * w = (xh + xm + xl) * (wh + wm + wl) = nine products of 8-bit significands, each EXACT in fp32,
// i.e. nine v_mfma_f32_16x16x32_bf16 (9 x 16 clocks) contract what eight fp32 MFMAs (8 x 32 clocks) do, with no narrow-precision input
// anywhere: what differs from the fp32 instruction is the order in which exact products meet in the fp32 accumulator (small terms first), as
// any tiling changes it.  SURVEY App. C forbids LOSSY narrow inputs; this is the lossless split of k_stream_q8b (activations) and k_gemm_b9
// (both sides).
//
// Structure: k_stream_q8b's sixteen EQUAL waves on LDS-DMA rings (kernels_stream_q8b.h), re-cut after the first build's timeline
// (profiles/r06_stream_b9_probe.txt: with one barrier per chunk all waves read, split and multiply in the SAME phases, so the matrix pipe
// idled while every wave waited for its operands and split its weights - 2460 clocks per chunk where its MFMAs need 1230):
//   * one workgroup per CU owns a contiguous block of <= MAXT 16-row weight tiles and ALL token columns: every weight byte is read once, no
//     partial sum leaves the chip (K-split launches excepted: wo / w2, k_stream_reduce_norm);
//   * chunks of 64 columns; TWO rings: NW weight images [MAXT * 16 rows][64 floats] and NX plane images [3][NCT * 16 rows][64 bf16].  The
//     weight ring runs ONE CHUNK AHEAD of the planes: in period ch (behind barrier ch) a wave multiplies chunk ch out of registers - its
//     weight fragment was read and split during period ch - 1 - and meanwhile reads and splits its fragment of chunk ch + 1, so no LDS
//     latency and no vector work stands between a barrier and the first MFMA behind it except the plane reads.  The planes come out of L2
//     (every workgroup reads all of X) and need a shallow ring, the weights come from HBM and get the rest of the 160 KB;
//   * waves 0..7 issue the weight DMAs, waves 8..15 the plane DMAs (`buffer_load_dwordx4 ... lds`, inline asm, counted s_waitcnt vmcnt):
//     the counter retires in order, so a wave that mixed both would have to drain its weight chunks in flight to see its planes land;
//   * X arrives as three bf16 planes (written by the kernel that produces the rows: k_rmsnorm_rows_s3, the attention kernels, the SiLU
//     epilogue, k_stream_reduce_norm) - never converted on the matmul side;
//   * W stays fp32 in HBM and in LDS and is split on the operand-read side: wave w owns k-block w & 1 (32 columns) of every chunk for tile
//     w / (2 CS) and the column tiles ((w / 2) % CS) NCT / CS + c: 9 MFMAs per (fragment, column tile), transposed (A = activations, B =
//     weights: a lane's four results belong to ONE weight row).  CS > 1 (few tiles per workgroup: wq|wk|wv, the K-split wo / w2) deals a
//     tile's columns to several waves, so that every SIMD gets the same number of MFMAs; those waves split the same fragment (5.5 vector
//     instructions per weight, cheap next to 9 MFMAs per 8 weights and column tile);
//   * the two k-block partial tiles of a tile meet in LDS and are added in wave order (stream_epilogue: bit-reproducible), then the launch's
//     epilogue: + residual | silu(w1 h) * (w3 h) on (w1, w3) tile pairs | RoPE + cache append - and optionally the result split into planes
//     for the next launch.
//   weight image: dense (the DMA writes lane-linearly); 16-byte granule g of row r at position g ^ (r & 15): a ds_read_b128 lane group (eight
//     rows at slot s, eight at slot s ^ 1) touches sixteen different positions;  plane image: 128-byte rows, granule g of row r at
//     g ^ ((r >> 1) & 7) (two rows per bank line).
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

// MAXT: 16-row weight tiles per workgroup; NCT: 16-token column tiles; CS: column parts (waves per tile and k-block); NW / NX: weight / plane images;
// NPROD: 9 (exact) - probe builds: 8 drops xl * wl, 6 also xm * wl and xl * wm (tools/b9s_probe: what each costs against an f64 product)
constexpr int B9S_TH = 1024, B9S_KC = 64;
__host__ __device__ constexpr size_t stream_b9_w_bytes(int maxt) { return (size_t)maxt * 16 * B9S_KC * 4; }
__host__ __device__ constexpr size_t stream_b9_x_bytes(int nct) { return (size_t)3 * nct * 16 * B9S_KC * 2; }
__host__ __device__ constexpr size_t stream_b9_lds_bytes(int maxt, int nct, int nw, int nx) { return nw * stream_b9_w_bytes(maxt) + nx * stream_b9_x_bytes(nct) + 1024; }
// column parts for a launch shape: as many as keep one tile per wave (TG = 8 / CS >= MAXT) and divide the column tiles
__host__ __device__ constexpr int stream_b9_cs(int maxt, int nct) {
    int cs = maxt <= 1 ? 8 : (maxt <= 2 ? 4 : (maxt <= 4 ? 2 : 1));
    while (cs > 1 && nct % cs) cs >>= 1;
    return cs;
}
// weight images that fit next to nx plane images (<= cap)
__host__ __device__ constexpr int stream_b9_nw(int maxt, int nct, int nx, int cap) {
    const long room = 160 * 1024 - 1024 - (long)nx * (long)stream_b9_x_bytes(nct);
    const int n = room <= 0 ? 0 : (int)(room / (long)stream_b9_w_bytes(maxt));
    return n < cap ? n : cap;
}
template <int MAXT, int NCT, int CS, int NW, int NX, bool XA, int NPROD = 9>
__global__ __launch_bounds__(B9S_TH) void k_stream_b9(const StreamArgs a) {
    constexpr int KC = B9S_KC, NWV = B9S_TH / 64, NB = 2;
    static_assert(NW >= 2 && NW <= 8 && NX >= 2 && NX <= 4, "rings");
    static_assert(NPROD == 9 || NPROD == 8 || NPROD == 6, "products");
    static_assert(CS == 1 || CS == 2 || CS == 4 || CS == 8, "column parts");
    static_assert(NCT % CS == 0, "column tiles per part");
    constexpr int TG = NWV / (NB * CS);         // tile groups = tiles a workgroup can hold
    static_assert(MAXT <= TG, "one tile per wave");
    constexpr int NCW = NCT / CS;               // column tiles per wave
    constexpr int XR = NCT * 16;                // staged activation rows per plane
    constexpr int GRW = 16, RPW = 4;            // 16-byte granules per weight row; weight rows per DMA instruction
    constexpr int GRX = 8, RPX = 8;             // granules per plane row; plane rows per DMA instruction
    constexpr int NLW = NWV / 2;                // waves 0..NLW-1 fetch weights, the others planes
    constexpr int NWI = MAXT * 16 / RPW, NXP = XR / RPX, NXI = 3 * NXP;
    constexpr int NIW_W = (NWI + NLW - 1) / NLW, NIW_X = (NXI + NLW - 1) / NLW, NIW = NIW_W > NIW_X ? NIW_W : NIW_X;
    constexpr uint32_t W_BYTES = MAXT * 16 * KC * 4, XP_BYTES = XR * KC * 2, X_BYTES = 3 * XP_BYTES, X0 = NW * W_BYTES, DUMMY = X0 + NX * X_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Q8B_TR_DECL;
    Q8B_STAMP(0);
    const uint32_t tiles_per_mat = a.M >> 4, T = tiles_per_mat * a.groups;
    const bool pairs = a.epi == ST_EPI_SILU_MUL;
    const uint32_t units = pairs ? tiles_per_mat : T, um = pairs ? 2u : 1u;
    const uint32_t S = a.ksplit > 1 ? a.ksplit : 1u, bg = (uint32_t)blockIdx.x / S, ks = (uint32_t)blockIdx.x - bg * S, ng = (uint32_t)gridDim.x / S;
    if (bg >= ng) return;
    const uint32_t t0 = um * (bg * units / ng), t1 = um * ((bg + 1) * units / ng);
    if (t1 <= t0) return;
    const uint32_t nt = t1 - t0;
    const uint32_t nch_all = a.K / KC, ch0 = ks * nch_all / S;
    const uint32_t nch = (ks + 1) * nch_all / S - ch0;
    const uint32_t kbase = ch0 * KC;
    if (nch == 0) return;
    const uint32_t r16 = (uint32_t)lane & 15, slot = (uint32_t)lane >> 4;
    auto xswz = [](uint32_t r) -> uint32_t { return (r >> 1) & 7u; };
    const bool wloader = wave < NLW;
    // ---- this wave's DMA instructions of a chunk: weight waves piece q = NLW j + wave of the tiles' rows, plane waves piece q = NLW j + wave - NLW of
    // the three planes.  Slots past the pieces issue with ONE active lane into a spare granule (a counted instruction that moves 16 bytes).
    const char* base[NIW];
    uint32_t voff[NIW], doff[NIW];
    bool real[NIW];
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const uint32_t q = (uint32_t)j * NLW + (uint32_t)(wloader ? wave : wave - NLW);
        if (wloader) {
            real[j] = j < NIW_W && q < (uint32_t)NWI;
            const uint32_t qq = real[j] ? q : 0u;
            const uint32_t rr = qq * RPW + (uint32_t)lane / GRW, gd = (uint32_t)lane % GRW, gs = gd ^ (rr & 15u);
            uint32_t ts = (qq * RPW) >> 4;                                 // tile slot of the instruction: uniform
            ts = ts < nt ? ts : nt - 1;                                    // slots beyond the block: a valid tile again, its sums are never stored
            const uint32_t v = t0 + ts;
            uint32_t g, tile;
            if (pairs) { g = v & 1u; tile = v >> 1; }
            else { g = (v >= tiles_per_mat ? 1u : 0u) + (v >= 2 * tiles_per_mat ? 1u : 0u); tile = v - g * tiles_per_mat; }   // (<= 3 matrices: no division)
            base[j] = (const char*)((g == 0 ? a.w[0] : (g == 1 ? a.w[1] : a.w[2])) + (size_t)tile * 16 * a.K + kbase);
            voff[j] = ((rr & 15u) * a.K + gs * 4u) * 4u;
            doff[j] = real[j] ? qq * 1024u : DUMMY;
        } else {
            real[j] = j < NIW_X && q < (uint32_t)NXI;
            const uint32_t qq = real[j] ? q : 0u;
            const uint32_t p = qq / NXP, xi = qq - p * NXP;
            const uint32_t rr = xi * RPX + (uint32_t)lane / GRX, gd = (uint32_t)lane % GRX, gs = gd ^ xswz(rr);
            const uint32_t c = rr < a.n ? rr : a.n - 1;                    // rows past the batch: the last row again (never stored)
            base[j] = (const char*)(a.xs + (size_t)p * a.xs_plane + kbase);
            voff[j] = (c * a.ldxs + gs * 8u) * 2u;
            doff[j] = real[j] ? p * XP_BYTES + xi * 1024u : DUMMY;
        }
    }
    typedef int i4v __attribute__((ext_vector_type(4)));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    // chunk `ch` of this wave's kind into its ring image (the DMA is inline asm ON PURPOSE - kernels_stream_q8b.h: through the builtin the compiler
    // drains the ring in front of every operand read)
    auto issue = [&](uint32_t ch) {
        const uint32_t im = wloader ? lds0 + (ch % NW) * W_BYTES : lds0 + X0 + (ch % NX) * X_BYTES;
        const uint32_t so = ch * (wloader ? (uint32_t)KC * 4u : (uint32_t)KC * 2u);
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            if (j >= (wloader ? NIW_W : NIW_X)) continue;
            const uint64_t b = (uint64_t)sgpr_ptr(base[j]);
            const i4v rs = {(int)(uint32_t)b, (int)((uint32_t)(b >> 32) & 0xffffu), 0x7fffffff, 0x00020000};   // raw buffer, stride 0 (stream_rsrc's words)
            const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(real[j] ? im + doff[j] : lds0 + DUMMY)), sov = (uint32_t)__builtin_amdgcn_readfirstlane((int)so);
            if (!real[j]) {
                unsigned long long saved;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(m0v), "v"(voff[j]), "s"(rs), "s"(sov) : "memory", "m0");
            } else if (wloader) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds" :: "s"(m0v), "v"(voff[j]), "s"(rs), "s"(sov) : "memory", "m0");   // weights: read once
            else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(voff[j]), "s"(rs), "s"(sov) : "memory", "m0");                    // planes: out of L2
        }
    };
    if (wloader) {
#pragma unroll
        for (int c = 0; c < NW; ++c) if ((uint32_t)c < nch) issue((uint32_t)c);
    } else {
#pragma unroll
        for (int c = 0; c < NX - 1; ++c) if ((uint32_t)c < nch) issue((uint32_t)c);
    }
    // ---- this wave's share of a chunk: k-block kb of tile tg for the column tiles cb .. cb + NCW - 1
    const uint32_t kb = (uint32_t)wave % NB, cpart = ((uint32_t)wave / NB) % CS, tg = (uint32_t)wave / (NB * CS), cb = cpart * NCW;
    const bool busy = tg < nt;                  // (a wave whose tile lies past the block has nothing to multiply: it only moves data)
    f4m acc[NCW];
#pragma unroll
    for (int c = 0; c < NCW; ++c) acc[c] = f4m{0.f, 0.f, 0.f, 0.f};
    const uint32_t woff = ((tg < (uint32_t)MAXT ? tg : (uint32_t)MAXT - 1) * 16 + r16) * (KC * 4) + (((kb * 8 + slot * 2) ^ r16) * 16);   // the fragment's first granule; the second at position ^ 1
    uint32_t xoff[NCW];
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
        const uint32_t row = (cb + (uint32_t)c) * 16 + r16;
        xoff[c] = X0 + (row * GRX + ((kb * 4 + slot) ^ xswz(row))) * 16;
    }
    // XA: the planes' ring runs one chunk ahead like the weights' (needs NX >= 3): a wave then holds the operands of its first column group of
    // chunk ch + 1 across barrier ch + 1 and the first MFMA behind a barrier waits for nothing
    constexpr int XD = XA ? 1 : 0;
    static_assert(!XA || NX >= 3, "planes one chunk ahead: three images");
    constexpr int WAIT_W0 = (NW - 1) * NIW_W < 64 ? (NW - 1) * NIW_W : 63, WAIT_W = (NW - 2) * NIW_W < 64 ? (NW - 2) * NIW_W : 63;
    constexpr int WAIT_X0 = (NX - 2) * NIW_X < 64 ? (NX - 2) * NIW_X : 63, WAIT_X = (NX - 2 - XD) * NIW_X < 64 ? (NX - 2 - XD) * NIW_X : 63;
    // column groups: the MFMA chains of a group's CG column tiles interleave; while group g multiplies out of one operand set the next group's
    // operands (the next chunk's first group behind the last) are read into the other
    constexpr int CG = NCW <= 2 ? NCW : (NCW == 3 ? (XA ? 1 : 3) : 2), NG = NCW / CG;   // (two sets of three column tiles' planes would not fit the registers)
    static_assert(NCW % CG == 0, "column groups");
    u4 wc[3], wn[3];                            // this chunk's weight fragment as three bf16 pieces, the next chunk's
    u4 xo[2][CG][3];                            // two operand sets of a column group's three planes
    auto read_x = [&](u4 (&o)[CG][3], uint32_t ch, int g) {
        const char* xim = smem_raw + (size_t)(ch % NX) * X_BYTES;
#pragma unroll
        for (int cc = 0; cc < CG; ++cc)
#pragma unroll
            for (int p = 0; p < 3; ++p) o[cc][p] = *(const u4*)(xim + xoff[g * CG + cc] + (size_t)p * XP_BYTES);
    };
    // products, small terms first: (x piece, w piece) with 0 = hi, 1 = mid, 2 = lo
#ifndef B9S_ABLATE
#define B9S_ABLATE 0   // tools/b9s_probe timing-only builds: 1 no plane DMAs behind the prologue | 2 no weight split | 4 one product | 8 no weight DMAs behind the prologue
#endif
    auto mfmas = [&](const u4 (&o)[CG][3], int g) {
        constexpr int PX[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0}, PW[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0};
        constexpr int P0 = (B9S_ABLATE & 4) ? 8 : (NPROD == 9 ? 0 : (NPROD == 8 ? 1 : 3));
#pragma unroll
        for (int q = P0; q < 9; ++q)
#pragma unroll
            for (int cc = 0; cc < CG; ++cc)
                acc[g * CG + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, o[cc][PX[q]]), __builtin_bit_cast(bf16x8, wc[PW[q]]), acc[g * CG + cc], 0, 0, 0);
    };
    // chunk 0's weights (and, XA, planes)
    if (wloader) { if ((uint32_t)NW <= nch) wait_vm<WAIT_W0>(); else wait_vm<0>(); }
    else if (XA) { if ((uint32_t)(NX - 1) <= nch) wait_vm<WAIT_X0>(); else wait_vm<0>(); }
    barrier_lds_only();
    if (busy) {
        const f4 w0 = *(const f4*)(smem_raw + woff), w1 = *(const f4*)(smem_raw + (woff ^ 16u));
        split3x8(w0, w1, &wc[0], &wc[1], &wc[2]);
        if (XA) read_x(xo[0], 0, 0);
    }
    for (uint32_t ch = 0; ch < nch; ++ch) {
        Q8B_LAP_START();
        // weight waves: chunk ch + 1 has landed (the NW - 2 younger ones may be in flight); plane waves: chunk ch (XA: ch + 1) has
        if (wloader) { if (ch + NW <= nch) wait_vm<WAIT_W>(); else wait_vm<0>(); }
        else { if (ch + NX - 1 <= nch) wait_vm<WAIT_X>(); else wait_vm<0>(); }
        Q8B_LAP(4);
        barrier_lds_only();                     // barrier ch: planes ch (+ 1) and weights ch + 1 are in their images; everybody has left planes ch - 1 and weights ch
        Q8B_LAP(5);
        Q8B_STAMP(2);
        const bool more = ch + 1 < nch;
        if (busy) {
            const char* wim = smem_raw + (size_t)((ch + 1) % NW) * W_BYTES;
            f4 w0 = f4{0.f, 0.f, 0.f, 0.f}, w1 = w0;
            // (LDS returns in order: the weight fragment first - the split in the shadow of the first MFMAs needs it - then the operands of later steps)
            if (more) { w0 = *(const f4*)(wim + woff); w1 = *(const f4*)(wim + (woff ^ 16u)); }
            if (!XA) read_x(xo[0], ch, 0);
            if (NG > 1) read_x(xo[1], ch, 1);
            else if (XA && more) read_x(xo[1], ch + 1, 0);
            __builtin_amdgcn_sched_barrier(0);   // the reads above are ISSUED in front of the MFMAs they hide behind
            mfmas(xo[0], 0);
            if constexpr ((B9S_ABLATE & 2) != 0) { wn[0] = __builtin_bit_cast(u4, w0); wn[1] = __builtin_bit_cast(u4, w1); wn[2] = wn[0]; }
            else split3x8(w0, w1, &wn[0], &wn[1], &wn[2]);   // (independent of the MFMAs around it: the scheduler interleaves)
            __builtin_amdgcn_sched_barrier(0);
        }
        // the refill of the images the barrier freed: behind the first MFMAs, not in front of them
        if (wloader) { if (ch + NW < nch && !(B9S_ABLATE & 8)) issue(ch + NW); }
        else { if (ch + NX - 1 < nch && !(B9S_ABLATE & 1)) issue(ch + NX - 1); }
        if (busy) {
#pragma unroll
            for (int g = 1; g < NG; ++g) {
                // operands of the step behind this one into the set the step in front of it has left
                if (g + 1 < NG) read_x(xo[(g + 1) & 1], ch, g + 1);
                else if (XA && more) read_x(xo[NG & 1], ch + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(xo[g & 1], g);
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((NG & 1) && XA) {   // an odd number of steps per chunk: the next chunk's first operands sit in the other set
#pragma unroll
                for (int cc = 0; cc < CG; ++cc)
#pragma unroll
                    for (int p = 0; p < 3; ++p) xo[0][cc][p] = xo[1][cc][p];
            }
            wc[0] = wn[0]; wc[1] = wn[1]; wc[2] = wn[2];
        }
        Q8B_LAP(1);
    }
    Q8B_STAMP(3);
    __syncthreads();   // the images are dead (every DMA was waited for: the last periods wait with vmcnt(0))
    Q8B_STAMP(6);
    stream_epilogue<MAXT, NCT, CS, true, NB>(a, smem_raw, (uint32_t)(DUMMY / 4), nullptr, t0, nt, ks, tiles_per_mat, [&](int t, int c) { return acc[c]; });
    Q8B_STAMP(7);
    Q8B_TR_STORE(NWV);
}

}  // namespace lh
