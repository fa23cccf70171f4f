// csrc/kernels_stream_q8b.h — block-int8 weights on the bf16 matrix pipe WITHOUT losing a bit (round 5; BASELINE config 4).
//
// Reference: the matmul is ComputeForwardMulMatFP32 (pkg/ml/ml.go:1976-2098); the quantised dtype is ours (ml.go:85-94, 123-124 only
// carry the enum and size tables; format and checker semantics: kernels_q8.h).
//
// Why: k_stream_q8 (kernels_stream.h) dequantises every weight to fp32 on the vector ALU (1.5 instructions per weight) and multiplies on
// v_mfma_f32_16x16x4_f32, which runs at the fp32 VECTOR rate (32 clocks per SIMD for 16 x 16 x 4) and shares the vector hardware with the
// conversion: 4x the MACs per byte of the fp32 model made every int8 launch from five rows on matrix-pipe-bound (profiles/r04_q8_stream_kernel.txt:
// data movement 22 us, + MFMAs 35 us, + conversion 48-51 us for w1|w3 of 7B).  Here the contraction runs on v_mfma_f32_16x16x32_bf16
// (16 clocks per SIMD for 16 x 16 x 32: 16x the rate) and stays EXACT:
//   * a quant q (|q| <= 127) is exactly a bf16 (8 significand bits);
//   * an fp32 activation is exactly the sum of three bf16: x = hi + mid + lo, each piece 8 consecutive bits of x's 24-bit significand
//     (hi = x with the low 16 bits cleared, r = x - hi exact, mid = r with the low 16 bits cleared, lo = r - mid exact: split3 below);
//   * so every product q * piece is exact in fp32 (8 + 8 bits) and  sum_k q_k x_k  over one quant block (K = 32 = ONE MFMA per piece)
//     = three MFMAs accumulating in fp32 - no narrow-precision input anywhere (SURVEY App. C forbids LOSSY narrow inputs);
//   * the block scale multiplies the BLOCK SUM:  acc += d_block * (sum_k q_k x_k)  - one fma per output element and block.  Against the
//     checker's  sum_k fl32(d q_k) x_k  this drops the rounding of d * q (the result is closer to the exact product of the stored
//     numbers); measured on 7B logits in tests/ and bench.py's int8 parity object.
// What it costs: the vector ALU still converts int8 -> bf16 (v_cvt_f32_i32 with a byte select, exact; v_perm_b32 packs two high
// halves: 1.5 instructions per weight, as before) - but the MFMAs behind them are 3 x 16 clocks per (16 rows x 32 columns x 16 tokens)
// instead of 8 x 32.
//
// Structure (what the measurements of round 5 asked for, profiles/r05_q8b_*.txt):
//   * a wave issues one vector instruction every ~5.5 clocks whatever the instruction, the SIMD executes four waves' instructions in that
//     time (tools/valu_rate_probe) - so the conversion must be spread over as many waves as the CU holds.  k_stream_q8's shape (four
//     loader waves + four MFMA waves, ONE converting wave per SIMD) was bound by that single wave's issue rate; its timeline with the
//     bf16 MFMAs showed the MFMA waves computing for 16.3 of the loop's 18.8 us.  Here the workgroup is SIXTEEN EQUAL waves (four per
//     SIMD): every wave issues a few of the chunk's LDS-DMA instructions (`buffer_load_dwordx4 ... lds`: raw bytes, nothing converted on
//     the way), then converts and multiplies its share - quant block w % NB of the chunk for the tiles t = w / NB (mod 16 / NB).
//   * ring of NIMG images, one workgroup barrier per chunk; partial tiles of the NB block-waves added in wave order (stream_epilogue).
//   * the MFMA is issued TRANSPOSED - A = activations (M = token), B = weights (N = weight row) - so a lane's four results belong to ONE
//     weight row (lane & 15) and need one scale.
//   image per chunk of KC columns:
//     weights  [MAXT * 16 rows][KC bytes]; 16-byte granule g of row r at position g ^ swz(r) (source-side swizzle: the ds_read_b64 of a
//              quant block - 16 rows x 2 slots per lane group - touches all 64 banks once)
//     scales   [MAXT tiles][16 rows][KC / 32] floats, no padding: one DMA instruction per tile with only the first 4 KC / 32 lanes active (EXEC
//              masked around it: an LDS-DMA writes 16 bytes per ACTIVE lane).  (4-byte DMAs that gather the scales transposed - conflict-free
//              reads - measured slower: twice the instructions, w2 at 8 rows 16.9 -> 19.3 us, profiles/r05_q8b_probe.txt)
//     x planes [3][XR rows][KC bf16]; granule g of row r at g ^ (r & 15) (g ^ 2 r with XR = 8); XR = 8 (up to eight token rows) or NCT * 16
// (Measured and dropped, profiles/r05_q8b_probe.txt: the converting waves fetching their activation pieces and scales straight from global
// memory into registers, one chunk ahead, with a weights-only ring of up to eight images - w1|w3 of 7B at 8 rows 36.0 us against 27.5:
// a wave-load of 16 token rows x 64 bytes is sixteen separate cache lines, and a deeper ring bought nothing, 32.2 us with two images.)
#pragma once
#include "kernels_stream.h"

namespace lh {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Activation rows -> planes (stand-alone form; the product fuses the split into the kernel that produces the rows).
// One workgroup per row; planes[p][row][K].
struct Split3Args {
    const float* x;      // [n][ldx]
    uint16_t* xs;        // planes
    uint64_t plane;      // elements between planes
    uint32_t K, ldx, ldxs;
};
__global__ __launch_bounds__(256) void k_split3_rows(const Split3Args a) {
    const float* xr = a.x + (size_t)blockIdx.x * a.ldx;
    for (uint32_t i = threadIdx.x; i < a.K / 4; i += 256) {
        const f4 v = ((const f4*)xr)[i];
        uint32_t h[4], m[4], l[4];
        split3(v.x, &h[0], &m[0], &l[0]); split3(v.y, &h[1], &m[1], &l[1]); split3(v.z, &h[2], &m[2], &l[2]); split3(v.w, &h[3], &m[3], &l[3]);
        uint16_t* o = a.xs + (size_t)blockIdx.x * a.ldxs + (size_t)i * 4;
        *(uint2*)(o) = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
        *(uint2*)(o + a.plane) = uint2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
        *(uint2*)(o + 2 * a.plane) = uint2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
    }
}

// RMSNorm * gamma of activation rows straight into the planes (k_rmsnorm_rows' arithmetic: fp32 squares, f64 sum, one fp32 scale, two
// roundings per element - ml.go:1753-1812, 1877-1914); y != nullptr: the fp32 rows as well.  One workgroup per row, rows of up to 8192 floats.
__global__ __launch_bounds__(256) void k_rmsnorm_rows_s3(const float* __restrict__ x, const float* __restrict__ gamma, float* __restrict__ y, uint16_t* __restrict__ xs, uint64_t plane, uint32_t d) {
    __shared__ double sred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xr = x + (size_t)blockIdx.x * d;
    constexpr int NV = 8;
    const uint32_t d4 = d / 4;
    f4 v[NV], g[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const uint32_t i = (uint32_t)tid + (uint32_t)j * 256;
        v[j] = ((const f4*)xr)[i < d4 ? i : 0];
        g[j] = ((const f4*)gamma)[i < d4 ? i : 0];
    }
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j)
        if ((uint32_t)tid + (uint32_t)j * 256 < d4) {
            s += (double)__fmul_rn(v[j].x, v[j].x); s += (double)__fmul_rn(v[j].y, v[j].y);
            s += (double)__fmul_rn(v[j].z, v[j].z); s += (double)__fmul_rn(v[j].w, v[j].w);
        }
    s = wave_sum_f64(s);
    if (lane == 0) sred[wave] = s;
    __syncthreads();
    const double mean = (((sred[0] + sred[1]) + sred[2]) + sred[3]) / (double)d;
    const float scale = (float)(1.0 / sqrt(mean + 1e-5));
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const uint32_t i = (uint32_t)tid + (uint32_t)j * 256;
        if (i < d4) {
            f4 o;
            o.x = __fmul_rn(g[j].x, __fmul_rn(v[j].x, scale)); o.y = __fmul_rn(g[j].y, __fmul_rn(v[j].y, scale));
            o.z = __fmul_rn(g[j].z, __fmul_rn(v[j].z, scale)); o.w = __fmul_rn(g[j].w, __fmul_rn(v[j].w, scale));
            if (y) ((f4*)(y + (size_t)blockIdx.x * d))[i] = o;
            uint32_t hh[4], mm[4], ll[4];
            split3(o.x, &hh[0], &mm[0], &ll[0]); split3(o.y, &hh[1], &mm[1], &ll[1]); split3(o.z, &hh[2], &mm[2], &ll[2]); split3(o.w, &hh[3], &mm[3], &ll[3]);
            uint16_t* op = xs + (size_t)blockIdx.x * d + (size_t)i * 4;
            *(uint2*)(op) = uint2{hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16)};
            *(uint2*)(op + plane) = uint2{mm[0] | (mm[1] << 16), mm[2] | (mm[3] << 16)};
            *(uint2*)(op + 2 * plane) = uint2{ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16)};
        }
    }
}

__host__ __device__ constexpr size_t stream_q8b_image_bytes(int maxt, int xr, int kc) { return (size_t)maxt * 16 * kc + (size_t)maxt * (kc / 32) * 64 + (size_t)3 * xr * kc * 2; }
constexpr int stream_q8b_xr(int nct, uint32_t n) { return (nct == 1 && n <= 8) ? 8 : nct * 16; }

// tools/q8b_probe builds with -DQ8B_TRACE: a timeline of one workgroup's waves 0 and 15 in 100 MHz ticks (stamps 0 start, 2 first barrier
// passed, 3 loop end, 6 epilogue start, 7 end; laps 1 computing, 4 waiting for its DMAs, 5 at barriers + issuing).  Nothing of it exists in the
// product build.  (The timing-only ablation builds that found the compiler's vmcnt(0) - profiles/r05_q8b_ablation.txt - are in the git history.)
#ifdef Q8B_TRACE
#ifndef Q8B_TRACE_BLOCK
#define Q8B_TRACE_BLOCK (gridDim.x / 2)
#endif
#define Q8B_TR_DECL unsigned long long tstamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlap = 0; const bool trace_on = a.trace != nullptr && blockIdx.x == Q8B_TRACE_BLOCK; const unsigned long long tclk0 = __builtin_amdgcn_s_memtime()
#define Q8B_TR_NOW() (__builtin_amdgcn_sched_barrier(0), tlap = __builtin_amdgcn_s_memrealtime(), __builtin_amdgcn_sched_barrier(0), tlap)
#define Q8B_STAMP(i) do { if (trace_on && !tstamp[i]) tstamp[i] = Q8B_TR_NOW(); } while (0)
#define Q8B_LAP_START() unsigned long long tl0 = Q8B_TR_NOW()
#define Q8B_LAP(i) do { const unsigned long long tn = Q8B_TR_NOW(); tstamp[i] += tn - tl0; tl0 = tn; } while (0)
#define Q8B_TR_STORE(nwv) do { if (trace_on && lane == 0 && (wave == 0 || wave == (nwv) - 1)) { for (int i_ = 0; i_ < 8; ++i_) a.trace[(wave ? 8 : 0) + i_] = tstamp[i_]; if (wave == 0) a.trace[16] = __builtin_amdgcn_s_memtime() - tclk0; } } while (0)
#else
#define Q8B_TR_DECL
#define Q8B_STAMP(i)
#define Q8B_LAP_START()
#define Q8B_LAP(i)
#define Q8B_TR_STORE(nwv)
#endif

// MAXT: 16-row weight tiles per workgroup; NCT: 16-token column tiles; KC: columns per chunk; NIMG: images in the ring;
// XR: activation rows staged per plane (8: launches of up to eight token rows stage half a column tile; else NCT * 16)
constexpr int Q8B_TH = 1024;
template <int MAXT, int NCT, int KC, int NIMG, int XR, int TH = Q8B_TH>
__global__ __launch_bounds__(TH) void k_stream_q8b(const StreamArgs a) {
    LH_TOUCH_ARGS(a.w[0], a.r[2], a.epi, a.gamma, a.ys_plane, a.ldys);   // the argument block's lines behind one wait (kernels_common.h)
    static_assert(KC == 128 || KC == 256 || KC == 512, "chunk");
    static_assert(NIMG >= 2 && NIMG <= 4, "ring");   // (the product launches up to three)
    static_assert(XR == NCT * 16 || (NCT == 1 && XR == 8), "staged activation rows");
    constexpr int NWV = TH / 64;                // waves
    static_assert(NWV >= KC / 32 && NWV % (KC / 32) == 0, "every quant block of a chunk needs its wave(s)");
    constexpr int NB = KC / 32;                 // quant blocks per chunk: 4 / 8 / 16
    constexpr int TG = NWV / NB;                // tile groups: 4 / 2 / 1
    constexpr int TPW = (MAXT + TG - 1) / TG;   // tiles per wave
    constexpr int GRW = KC / 16;                // 16-byte granules per weight row
    constexpr int RPW = GRW >= 64 ? 1 : 64 / GRW;   // weight rows per DMA instruction: 8 / 4 / 2
    constexpr int GRX = KC / 8;                 // granules per plane row: 16 / 32 / 64
    constexpr int RPX = 64 / GRX;               // plane rows per DMA instruction: 4 / 2 / 1
    constexpr int NWI = MAXT * 16 / RPW, NSI = MAXT, NXP = XR / RPX, NXI = 3 * NXP, NI = NWI + NSI + NXI;
    constexpr int NIW = (NI + NWV - 1) / NWV;   // DMA instructions per wave and chunk
    constexpr int WAITN = NIW * (NIMG - 2) < 64 ? NIW * (NIMG - 2) : 63;
    constexpr uint32_t W_BYTES = MAXT * 16 * KC, S_BYTES = MAXT * NB * 64, XP_BYTES = XR * KC * 2, IMG_BYTES = W_BYTES + S_BYTES + 3 * XP_BYTES;
    constexpr int WSH = KC == 128 ? 1 : 0;      // swz(r) = (r >> WSH) & min(GRW - 1, 15): 128-byte rows alias every second row, longer ones every row
    constexpr uint32_t WMASK = GRW - 1 < 15 ? GRW - 1 : 15;
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
    // (32-bit arithmetic: the 64-bit divisions that stood here and a division per DMA piece below were half of the ~1100 instructions in front of
    //  the first DMA; removing them measured NOTHING - the 4.4 us to the first barrier are the first chunk's way from HBM, not this code)
    const uint32_t t0 = um * (bg * units / ng), t1 = um * ((bg + 1) * units / ng);
    if (t1 <= t0) return;
    const uint32_t nt = t1 - t0;
    const uint32_t nch_all = a.K / KC, ch0 = ks * nch_all / S;
    const uint32_t nch = (ks + 1) * nch_all / S - ch0;
    const uint32_t kbase = ch0 * KC;
    if (nch == 0) return;
    const uint32_t r16 = (uint32_t)lane & 15, slot = (uint32_t)lane >> 4;
    // plane-row swizzle: a ds_read_b128 lane group holds 16 rows at two neighbouring slots; with eight staged rows (each read by two
    // lanes) the rows spread over the even offsets so that the two slots' positions stay disjoint
    auto xswz = [](uint32_t r) -> uint32_t { return XR == 8 ? ((r & 7u) << 1) : (r & 15u); };
    // ---- this wave's DMA instructions of a chunk: piece q = NWV j + wave (weights, the tiles' scales, the three activation planes)
    const char* base[NIW];
    uint32_t voff[NIW], sstep[NIW], doff[NIW];
    bool narrow[NIW];                                                      // a scale piece (fewer active lanes)
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        uint32_t q = (uint32_t)j * NWV + (uint32_t)wave;
        q = q < (uint32_t)NI ? q : (uint32_t)NI - 1;                       // surplus slots repeat the last piece (same bytes to the same place)
        auto tile_base = [&](uint32_t ts, bool scales) -> const char* {
            ts = ts < nt ? ts : nt - 1;
            const uint32_t v = t0 + ts;
            uint32_t g, tile;
            if (pairs) { g = v & 1u; tile = v >> 1; }
            else { g = (v >= tiles_per_mat ? 1u : 0u) + (v >= 2 * tiles_per_mat ? 1u : 0u); tile = v - g * tiles_per_mat; }   // (<= 3 matrices: no division)
            if (scales) return (const char*)((g == 0 ? a.ws[0] : (g == 1 ? a.ws[1] : a.ws[2])) + (size_t)tile * 16 * (a.K / 32) + kbase / 32);
            return (const char*)(g == 0 ? a.w[0] : (g == 1 ? a.w[1] : a.w[2])) + (size_t)tile * 16 * a.K + kbase;
        };
        narrow[j] = q >= (uint32_t)NWI && q < (uint32_t)(NWI + NSI);
        if (q < (uint32_t)NWI) {
            const uint32_t rr = q * RPW + (uint32_t)lane / GRW, gd = (uint32_t)lane % GRW, gs = gd ^ ((rr >> WSH) & WMASK);
            base[j] = tile_base((q * RPW) >> 4, false);
            voff[j] = (rr & 15u) * a.K + gs * 16u;
            sstep[j] = KC;
            doff[j] = q * 1024u;
        } else if (q < (uint32_t)(NWI + NSI)) {
            const uint32_t ts = q - NWI;
            constexpr uint32_t LPR = KC / 128;                             // 16-byte pieces per row's scales: 1 / 2 / 4
            const uint32_t l = (uint32_t)lane & (16u * LPR - 1u);          // (lanes past 16 LPR are masked off when the instruction issues)
            base[j] = tile_base(ts, true);
            voff[j] = ((l / LPR) * (a.K / 32) + (l % LPR) * 4u) * 4u;
            sstep[j] = (KC / 32) * 4;
            doff[j] = W_BYTES + ts * (NB * 64u);
        } else {
            const uint32_t xq = q - NWI - NSI, p = xq / NXP, xi = xq - p * NXP;
            const uint32_t rr = xi * RPX + (uint32_t)lane / GRX, gd = (uint32_t)lane % GRX, gs = gd ^ xswz(rr);
            const uint32_t c = rr < a.n ? rr : a.n - 1;
            base[j] = (const char*)(a.xs + (size_t)p * a.xs_plane + kbase);
            voff[j] = (c * a.ldxs + gs * 8u) * 2u;
            sstep[j] = KC * 2;
            doff[j] = W_BYTES + S_BYTES + p * XP_BYTES + xi * 1024u;
        }
    }
    // The DMA is written as inline asm ON PURPOSE: through __builtin_amdgcn_raw_ptr_buffer_load_lds the compiler sees a store to LDS and puts
    // s_waitcnt vmcnt(0) in front of this wave's next operand read of the ring (it cannot know that the images differ) - every wave then
    // drains ALL its chunks in flight once per chunk and the launch takes the SUM of its HBM time and its compute time (seen in the ISA and
    // in the timeline, profiles/r05_q8b_ablation.txt).  The waits for these DMAs are the explicit counted ones of the loop.
    typedef int i4v __attribute__((ext_vector_type(4)));
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem_raw;
    auto issue = [&](uint32_t ch) {
        const uint32_t cc = ch;
        const uint32_t im = lds0 + (ch % NIMG) * IMG_BYTES;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const uint64_t b = (uint64_t)sgpr_ptr(base[j]);
            const i4v rs = {(int)(uint32_t)b, (int)((uint32_t)(b >> 32) & 0xffffu), 0x7fffffff, 0x00020000};   // raw buffer, stride 0 (stream_rsrc's words)
            const uint32_t m0v = (uint32_t)__builtin_amdgcn_readfirstlane((int)(im + doff[j])), so = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cc * sstep[j]));
            if (narrow[j] && KC < 512) {   // a tile's scales: 16 KC / 128 active lanes
                unsigned long long saved;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %5\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b64 exec, %0"
                             : "=&s"(saved) : "s"(m0v), "v"(voff[j]), "s"(rs), "s"(so), "s"(KC == 128 ? 0xffffull : 0xffffffffull) : "memory", "m0");
            } else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(m0v), "v"(voff[j]), "s"(rs), "s"(so) : "memory", "m0");
        }
    };
#pragma unroll
    for (int c = 0; c < NIMG - 1; ++c) if ((uint32_t)c < nch) issue((uint32_t)c);
    // ---- this wave's share of a chunk: quant block kb for the tiles tg, tg + TG, ...
    const uint32_t kb = (uint32_t)wave % NB, tg = (uint32_t)wave / NB;
    f4m acc[TPW][NCT];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[j][c] = f4m{0.f, 0.f, 0.f, 0.f};
    uint32_t woff[TPW], soff[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        uint32_t t = tg + (uint32_t)j * TG;
        t = t < (uint32_t)MAXT ? t : (uint32_t)MAXT - 1;                   // (a slot past the tiles: the last one again, its sums are never stored)
        woff[j] = t * 16 * KC + r16 * KC + (((kb * 2 + (slot >> 1)) ^ ((r16 >> WSH) & WMASK)) * 16) + (slot & 1u) * 8;
        soff[j] = W_BYTES + t * (NB * 64) + r16 * (NB * 4) + kb * 4;
    }
    uint32_t xoff[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        const uint32_t row = XR == 8 ? (r16 & 7u) : (uint32_t)c * 16 + r16;   // (rows 8..15 of a half tile: copies of 0..7, never stored)
        xoff[c] = W_BYTES + S_BYTES + (row * GRX + ((kb * 4 + slot) ^ xswz(row))) * 16;
    }
    for (uint32_t ch = 0; ch < nch; ++ch) {
        const char* im = smem_raw + (size_t)(ch % NIMG) * IMG_BYTES;
        Q8B_LAP_START();
        // this wave's pieces of chunk ch have landed; the younger chunks' (min(NIMG - 2, chunks left) of them) may still be in flight
        {
            const uint32_t left = nch - 1 - ch;
            if (left >= (uint32_t)(NIMG - 2)) wait_vm<WAITN>();
            else if (NIMG >= 4 && left == 1) wait_vm<(NIW < 64 ? NIW : 63)>();
            else wait_vm<0>();
        }
        Q8B_LAP(4);
        barrier_lds_only();                     // barrier ch: every piece of chunk ch is in its image, and everybody has left chunk ch - 1's ...
        if (ch + NIMG - 1 < nch) issue(ch + NIMG - 1);   // ... which takes chunk ch + NIMG - 1
        Q8B_LAP(5);
        Q8B_STAMP(2);
        u4 xo[3][NCT];
        uint2 raw[TPW];
        float d[TPW];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int c = 0; c < NCT; ++c) xo[p][c] = *(const u4*)(im + xoff[c] + (size_t)p * XP_BYTES);
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            raw[j] = *(const uint2*)(im + woff[j]);
            d[j] = *(const float*)(im + soff[j]);
        }
        // int8 -> bf16: (float)q exact, its high half IS the bf16; v_perm_b32 packs two high halves
        u4 wb[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int d0 = (int)raw[j].x, d1 = (int)raw[j].y;
            const uint32_t f0 = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d0)), f1 = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d0 >> 8));
            const uint32_t f2 = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d0 >> 16)), f3 = __builtin_bit_cast(uint32_t, (float)(d0 >> 24));
            const uint32_t f4_ = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d1)), f5 = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d1 >> 8));
            const uint32_t f6 = __builtin_bit_cast(uint32_t, (float)(int)(signed char)(d1 >> 16)), f7 = __builtin_bit_cast(uint32_t, (float)(d1 >> 24));
            wb[j] = u4{__builtin_amdgcn_perm(f1, f0, 0x07060302u), __builtin_amdgcn_perm(f3, f2, 0x07060302u),
                       __builtin_amdgcn_perm(f5, f4_, 0x07060302u), __builtin_amdgcn_perm(f7, f6, 0x07060302u)};
        }
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            // block sums, small pieces first; the wave's tiles interleaved so that consecutive MFMAs are independent
            f4m ps[TPW];
#pragma unroll
            for (int j = 0; j < TPW; ++j) ps[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xo[2][c]), __builtin_bit_cast(bf16x8, wb[j]), f4m{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPW; ++j) ps[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xo[1][c]), __builtin_bit_cast(bf16x8, wb[j]), ps[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPW; ++j) ps[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, xo[0][c]), __builtin_bit_cast(bf16x8, wb[j]), ps[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                acc[j][c][0] = fmaf(d[j], ps[j][0], acc[j][c][0]); acc[j][c][1] = fmaf(d[j], ps[j][1], acc[j][c][1]);
                acc[j][c][2] = fmaf(d[j], ps[j][2], acc[j][c][2]); acc[j][c][3] = fmaf(d[j], ps[j][3], acc[j][c][3]);
            }
        }
        Q8B_LAP(1);
    }
    Q8B_STAMP(3);
    __syncthreads();   // the images are dead
    Q8B_STAMP(6);
    stream_epilogue<MAXT, NCT, 1, true, NB>(a, smem_raw, (uint32_t)((size_t)NIMG * IMG_BYTES / 4), nullptr, t0, nt, ks, tiles_per_mat, [&](int t, int c) { return acc[t / TG][c]; });
    Q8B_STAMP(7);
    Q8B_TR_STORE(NWV);
}

}  // namespace lh
