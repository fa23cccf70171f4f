// csrc/kernels_stream.h — weight-streaming GEMM for SHORT prompts of 9..64 token rows (fp32):  Y[n][M] = X[n][K] . W[M][K]^T (+ R).
//
// Reference: server.Do feeds the whole prompt as ONE Eval (pkg/server/server.go:185-192); the matmuls are ComputeForwardMulMatFP32
// (pkg/ml/ml.go:1976-2098).  Up to ~48 rows one pass over the 26.4 GB of 7B weights is HBM-bound (the fp32 matrix pipe needs 2.7 ms
// at 32 rows, the stream 4.2 ms), so the kernel is built around the weight stream, like the decode GEMV, and the matrix cores only
// have to keep up.  Round 2 ran this range on the prefill tile GEMM (k_gemm_glds, 64 x 128 tiles + split-K): its LDS image makes one
// load instruction touch 8 rows x 128 B and it reached 2.8 TB/s (profiles/r02b_ttft_kernel_trace.txt: 72.7 us per GEMM launch, four
// per layer, plus a split-K reduce pass each).
//   * grid = #CU workgroups of 4 waves; a workgroup owns a contiguous block of 16-row tiles of the (grouped) weight matrix and ALL
//     token columns, so every weight byte is read once and no partial sum leaves the chip (no split-K pass);
//   * weights travel global -> registers -> LDS in K-chunks of KC = 128 / 256 / 512 columns (the host picks the largest that fits the
//     registers and LDS: few-row matrices like wo and w2 need the long chunks to have enough HBM bytes in flight): a wave-instruction
//     loads 1 KB contiguous of ONE row from KC = 256 on (2 rows x 512 B at 128), non-temporal; two chunks stay in flight in registers
//     (the matrix work of the current chunk runs under them), then ds_write_b128 into a [rows][KC + 4] image (the pad spreads the 16
//     rows of a tile over all 64 banks for the operand reads);
//   * activations the same way into a [columns][KC + 4] image (they come out of L2: every workgroup reads all of X);
//   * v_mfma_f32_16x16x4_f32: A = 16 weight rows x 4 k, B = 4 k x 16 token columns.  A lane (row or column l & 15, slot l >> 4) takes
//     one ds_read_b128 = 4 consecutive k of its row / column and feeds them to 4 consecutive MFMAs (MFMA s contracts the k's
//     16 b + 4 slot + s: any grouping of k's is valid as long as A and B agree).  The KC / 16 k-blocks of a chunk are dealt to the 4 waves,
//     every wave runs ALL row tiles for its k-blocks: the B operands are read once per k-block and reused across the
//     tiles, consecutive MFMAs go to different accumulators (40-cycle dependent latency vs 32-cycle issue), and the waves are
//     balanced whatever the tile count;
//   * after the last chunk the four waves' partial tiles meet in LDS and are added in wave order (bit-reproducible), + residual.
// Summation order differs from the scalar reference (k interleaved over 4 waves x 4 slots): within 1e-4 like every MFMA path.
// (The kernel this paragraph was written for - k_stream_mm, every wave loading AND multiplying - left the product in round 6: nothing reached it
// any more; tools/kernels_stream_mm_r1.h keeps it for tools/stream_mm_check.  Image layout, operand mapping and summation structure live on in
// k_stream_mm2 below.)
#pragma once
#include "kernels_llama.h"

namespace lh {

struct StreamArgs {
    const float* w[3];   // matrices of the group, each [M][K] row-major
    float* y[3];         // outputs, y[g][c * ldy + row]
    const float* r[3];   // optional residuals, same layout as y
    const float* ws[3];  // block-int8 models (k_stream_q8b): w[] are the int8 planes [M][K], ws[] the fp32 scales [M][K / 32]
    const float* x;      // activations [n][K], row c at x + c * ldx
    uint32_t groups, M, K, n, ldx, ldy;
#ifdef Q8B_TRACE
    unsigned long long* trace;   // tools/q8b_probe: timeline stamps of one workgroup
#endif
    // fused epilogues of k_stream_mm2 (the launches llama.Eval makes, llama.go:255-297 and :346-361)
    uint32_t epi;        // ST_EPI_*
    float* q_out;        // ST_EPI_QKV_ROPE: roped Q [n][d]; the roped K rows and the V rows go to the cache at positions past + c
    float* k_cache;      //   this layer's slot [ctx][d]
    float* v_cache;
    const double2* rope; //   [pos][hd / 2] (cos, sin)
    uint32_t hd, past;
    // batched Eval (rows of DIFFERENT streams, lh_batch): column c is at position rows[c].pos of ITS OWN cache rows[c].kc / .vc (+ kv_off floats:
    // this layer's slot); past, k_cache and v_cache above are then unused
    const BatchRow* rows;
    uint64_t kv_off;
    // RMSNorm folded into the launch (k_stream_mm2): x is the RAW residual stream, gamma the norm weight [K].  The norm is linear in its
    // per-token scale s = fl32(1 / sqrt(mean(x^2) + 1e-5)), so the kernel contracts W with gamma * x and multiplies the sums by s in the
    // epilogue; s comes from the x chunks the loader waves stage anyway (fp32 squares, f64 sum, fixed order).  Rounding differs from
    // the reference's fl(gamma * fl(x * s)) by two fp32 roundings per term (ml.go:1788-1808, 1906) - inside the GEMM's own 1e-6 noise.
    const float* gamma;
    // K-split (k_stream_mm2, plain epilogue only): ksplit = S > 1 deals the workgroups in groups of S; group j owns a block of 16-row
    // tiles like a whole workgroup does otherwise (so S times as many rows), member s contracts K-chunks [s nch / S, (s + 1) nch / S) and
    // writes its partial sums to y[g] + s * ysplit (no residual); k_stream_reduce_norm adds the S partials in s order.  For the
    // single-tile matrices (wo, w2: M / 16 = #CU tiles): every workgroup reads all of X out of L2, n / 16 bytes per weight byte, which at
    // 17..32 rows is twice the weight stream; with S = 2 a workgroup stages half of X for two tiles.  (The later traffic probes,
    // profiles/r02d_stream_traffic_probe.txt, show the gain is in loader instructions per weight byte rather than in L2 traffic.)
    uint32_t ksplit;
    uint64_t ysplit;     // floats between the partial outputs
    uint32_t tiled;      // the matrices are stored chunk-major: [K / KC][M / 16][16][KC] (stream_tile_layout): a workgroup's rows of one
                         // K-chunk are ONE contiguous run, and so are all workgroups' together
    // k_stream_q8b (kernels_stream_q8b.h): the activations as three bf16 planes (x = hi + mid + lo exactly), plane p row c at
    // xs + p * xs_plane + c * ldxs (elements of 2 bytes); x above is then unused.  ys != nullptr: the epilogue ALSO writes its fp32 result
    // split the same way (the next launch's activations: silu*mul -> w2), row c at ys + p * ys_plane + c * ldys
    const uint16_t* xs;
    uint64_t xs_plane;
    uint32_t ldxs;
    uint16_t* ys;
    uint64_t ys_plane;
    uint32_t ldys;
};

enum { ST_EPI_STORE = 0, ST_EPI_SILU_MUL = 1, ST_EPI_QKV_ROPE = 2 };   // plain (+ residual) | y[0] = silu(w[0] x) * (w[1] x) | RoPE + cache append
constexpr int ST_TH = 256;   // threads; the K-chunk KC (floats per row and step) is a template parameter, LDS row pitch KC + 4 floats

__host__ __device__ inline size_t stream_lds_bytes(int maxt, int nct, int kc) { return (size_t)(maxt + nct) * 16 * (kc + 4) * 4; }

typedef float f4m __attribute__((ext_vector_type(4)));

// The fp32 loader waves of k_stream_mm2 fetch through buffer_load from two column tiles on (n > 16): a uniform resource per register
// (base = first row of its 16-row tile), a constant lane offset and the chunk offset in an SGPR, so no vector ALU instruction stands in
// front of a load.  Why: profiles/r02d_stream_traffic_probe.txt - from 17 rows on the launch is not memory-bound; the loads' 64-bit vector
// address adds wait while the MFMA wave of the same SIMD issues back to back, loading and computing alternate.  Measured round 3
// (profiles/r03_stream_buffer_loads.txt, w1|w3 / wq|wk|wv of 7B): 32 rows 88.6 -> 79.2 / 47.6 -> 43.8 us, 48 rows 121.3 -> 105.0 / 64.0 -> 58.8,
// 64 rows 143.3 -> 130.2 / 79.5 -> 73.7; with ONE column tile (<= 16 rows, HBM-bound) it measured 2-3 % slower and stays on global loads.
// Tried with it and dropped (same file): pacing the MFMA waves with s_nop (only slower from 32 rows on; +3 % at 16 rows on w1|w3, nothing
// on wq|wk|wv) and letting the loader waves sleep behind the chunk barrier (no consistent gain).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t stream_rsrc(const void* base) {
    // raw buffer (stride 0), no range limit in practice, 32-bit data format (gfx9 resource word 3)
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}

// s_waitcnt vmcnt(N) as an instruction the compiler's counter bookkeeping understands (gfx9 encoding)
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}

// Workgroup barrier that waits for this wave's LDS operations (operand reads of the image about to be overwritten) but NOT for its
// global loads: __syncthreads() would drain the chunks in flight.  The LDS wait is not optional: with one wave per SIMD a write another
// wave issues behind the barrier can overtake a read this wave issued in front of it but that has not executed yet (seen as wrong
// sums in some tiles with four column tiles, tests/test_gpu_ops.py::test_mul_mat_weights[11008-256-33]).
__device__ __forceinline__ void barrier_lds_only() {
    __builtin_amdgcn_s_waitcnt(0xF | (0x7 << 4) | (0x0 << 8) | (0x3 << 14));   // lgkmcnt(0), vmcnt and expcnt untouched
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// x = hi + mid + lo exactly, each piece a bf16 (its 16 bits returned): the activation format of k_stream_q8b (kernels_stream_q8b.h)
__device__ __forceinline__ void split3(float x, uint32_t* hi, uint32_t* mid, uint32_t* lo) {
    const uint32_t h = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
    const float r1 = __fsub_rn(x, __builtin_bit_cast(float, h));          // exact: the low 16 bits of x's significand
    const uint32_t m = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
    const float r2 = __fsub_rn(r1, __builtin_bit_cast(float, m));         // exact: <= 8 significant bits
    *hi = h >> 16; *mid = m >> 16; *lo = __builtin_bit_cast(uint32_t, r2) >> 16;
}
// four consecutive features `row..row + 3` of token row `col` into the three output planes (StreamArgs::ys)
__device__ __forceinline__ void stream_store_split3(const StreamArgs& a, uint32_t col, uint32_t row, f4 v) {
    uint32_t h[4], m[4], l[4];
    split3(v.x, &h[0], &m[0], &l[0]); split3(v.y, &h[1], &m[1], &l[1]); split3(v.z, &h[2], &m[2], &l[2]); split3(v.w, &h[3], &m[3], &l[3]);
    uint16_t* o = a.ys + (size_t)col * a.ldys + row;
    *(uint2*)(o) = uint2{h[0] | (h[1] << 16), h[2] | (h[3] << 16)};
    *(uint2*)(o + a.ys_plane) = uint2{m[0] | (m[1] << 16), m[2] | (m[3] << 16)};
    *(uint2*)(o + 2 * a.ys_plane) = uint2{l[0] | (l[1] << 16), l[2] | (l[3] << 16)};
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Epilogue of the wave-specialised kernels (k_stream_mm2, k_stream_dma): the partial tiles of the four MFMA waves (waves 4..7, each holding
// the sums over its share of every chunk's k-blocks) meet in LDS, thread (column, row quad) adds them in wave order (bit-reproducible) and
// applies the launch's epilogue: + residual | silu(w1 h) * (w3 h) on a (w1, w3) tile pair | RoPE + cache append.  lds_floats = floats of
// the dynamic LDS that may be overwritten (the images are dead: the caller has passed a workgroup barrier); scales = per-column factors of
// a folded RMSNorm (k_stream_mm2) or nullptr; acc_of(t, c) = this wave's partial tile.  CS = 2 (k_stream_dma, seven / eight column tiles):
// the MFMA waves are dealt as 2 K-groups x 2 column halves, MFMA wave w holds the sums of K-group w >> 1 for column tiles
// (w & 1) NCT / 2 + c; two partials per tile meet instead of four.
// TR (k_stream_q8b): the MFMA was issued transposed - a lane's four results are tokens 4 slot + i of ONE weight row (lane & 15).
// NB > 0 (k_stream_q8b, k_stream_b9: equal waves): wave w holds the sums over quant / k-block w % NB of every chunk for the column tiles
// ((w / NB) % CS) NCT / CS + c of the tiles t with t % (waves / (NB CS)) == w / (NB CS); NB partials per tile meet.
// TS = 2 (k_stream_b9, wave-specialised): MFMA wave w = 2 K-group + part holds the sums of its K-group for the tiles [0, ceil(MAXT / 2)) (part 0) or
// [ceil(MAXT / 2), MAXT) (part 1) and all column tiles; two partials per tile meet.
// TS = 3 (k_stream_b9 on 32 x 32 x 16 MFMAs): MFMA wave w holds K-group w of every tile in a layout of its own; acc_of(t, base) stores the wave's sums
// of tile t, element (token c, row r) at base[c * 16 + r]; four partials per tile meet.
template <int MAXT, int NCT, int CS = 1, bool TR = false, int NB = 0, int TS = 1, typename AccFn>
__device__ __forceinline__ void stream_epilogue(const StreamArgs& a, char* smem_raw, uint32_t lds_floats, const float* scales, uint32_t t0, uint32_t nt, uint32_t ks,
                                                uint32_t tiles_per_mat, AccFn&& acc_of) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t r16 = (uint32_t)lane & 15, slot = (uint32_t)lane >> 4;
    const bool pairs = a.epi == ST_EPI_SILU_MUL;
    auto tile_of = [&](uint32_t v, uint32_t* g, uint32_t* tile) {   // virtual tile -> (matrix, 16-row tile in it)
        if (pairs) { *g = v & 1u; *tile = v >> 1; }
        else { *g = (v >= tiles_per_mat ? 1u : 0u) + (v >= 2 * tiles_per_mat ? 1u : 0u); *tile = v - *g * tiles_per_mat; }   // (<= 3 matrices: no division)
    };
    constexpr int NC = NCT * 16, NKG = NB > 0 ? NB : (TS == 2 ? 2 : 4 / CS), NCW = NCT / CS;   // K-groups whose partial tiles meet; column tiles per MFMA wave
    const uint32_t TGX = NB > 0 ? ((uint32_t)blockDim.x >> 6) / (uint32_t)(NB > 0 ? NB * CS : 1) : 1u;   // tile groups of an equal-waves workgroup
    static_assert(CS == 1 || (NB == 0 && CS == 2 && NCT % 2 == 0) || (NB > 0 && NCT % CS == 0), "column split");
    static_assert(TS == 1 || ((TS == 2 || TS == 3) && NB == 0 && CS == 1), "tile split");
    constexpr uint32_t TS0 = (MAXT + 1) / 2;    // TS = 2: first tile of part 1
    float* part = (float*)smem_raw;
    constexpr uint32_t TILE_FLOATS = (uint32_t)NKG * NC * 16;
    const uint32_t batch = (lds_floats / TILE_FLOATS) & ~1u;   // even: a pair never straddles two batches
    // second phase: thread (column, row quad) of thread group `egrp` adds the partials of the tiles egrp, egrp + negrp, ... (tile pairs under
    // silu * mul) - every 4 NC threads of the workgroup take tiles of their own (round 5: one group did all of them while up to fifteen waves idled)
    const uint32_t egrp = (uint32_t)tid / (4u * NC), negrp = (uint32_t)blockDim.x / (4u * NC), etid = (uint32_t)tid - egrp * (4u * NC);
    const uint32_t col = etid >> 2, quad = etid & 3;
    const float nscale = (scales && col < (uint32_t)NC) ? scales[col] : 1.0f;
    const uint32_t kg = NB > 0 ? (uint32_t)wave % (uint32_t)NKG : ((CS == 2 || TS == 2) ? (uint32_t)(wave - 4) >> 1 : (uint32_t)(wave - 4));
    const uint32_t tpart = (uint32_t)(wave - 4) & 1u;
    const uint32_t cbase = NB > 0 ? (((uint32_t)wave / (uint32_t)NKG) % (uint32_t)CS) * NCW : (CS == 2 ? ((uint32_t)(wave - 4) & 1u) * NCW : 0u);
    const uint32_t tgw = NB > 0 ? (uint32_t)wave / (uint32_t)(NKG * CS > 0 ? NKG * CS : 1) : 0u;
    auto tile_sum = [&](uint32_t slot_in_batch) {
        const float* p = part + (size_t)slot_in_batch * TILE_FLOATS + (size_t)col * 16 + quad * 4;
        f4 s = *(const f4*)p;
#pragma unroll
        for (int w = 1; w < NKG; ++w) {
            const f4 q = *(const f4*)(p + (size_t)w * NC * 16);
            s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
        }
        if (scales) { s.x = __fmul_rn(s.x, nscale); s.y = __fmul_rn(s.y, nscale); s.z = __fmul_rn(s.z, nscale); s.w = __fmul_rn(s.w, nscale); }
        return s;
    };
    for (uint32_t tb = 0; tb < nt; tb += batch) {
        if (NB > 0 || wave >= 4) {
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                if constexpr (TS == 3) {
                    if ((uint32_t)t >= tb && (uint32_t)t < tb + batch && (uint32_t)t < nt) acc_of(t, part + (size_t)(t - tb) * TILE_FLOATS + (size_t)kg * NC * 16);
                } else
                if ((uint32_t)t >= tb && (uint32_t)t < tb + batch && (uint32_t)t < nt && (NB == 0 || (uint32_t)t % TGX == tgw) && (TS != 2 || ((uint32_t)t >= TS0) == (tpart != 0))) {
#pragma unroll
                    for (int c = 0; c < NCW; ++c) {
                        const f4m v = acc_of(t, c);
                        if constexpr (TR) {
                            float* pp = part + (size_t)(t - tb) * TILE_FLOATS + ((size_t)kg * NC + (cbase + c) * 16 + slot * 4) * 16 + r16;
                            pp[0] = v[0]; pp[16] = v[1]; pp[32] = v[2]; pp[48] = v[3];
                        } else
                        *(f4m*)(part + (size_t)(t - tb) * TILE_FLOATS + ((size_t)kg * NC + (cbase + c) * 16 + r16) * 16 + slot * 4) = v;
                    }
                }
            }
        }
        __syncthreads();
        if (egrp < negrp && col < a.n) {
            if (a.epi == ST_EPI_SILU_MUL) {
                for (uint32_t t = tb + 2 * egrp; t + 1 < tb + batch && t + 1 < nt; t += 2 * negrp) {   // (w1 tile, w3 tile) of the same rows
                    const f4 s1 = tile_sum(t - tb), s3 = tile_sum(t + 1 - tb);
                    const uint32_t row = ((t0 + t) >> 1) * 16 + quad * 4;
                    f4 o;   // Silu(w1 h) then Mul(., w3 h): ml.go:2587-2589, 1877-1914 (llama.go:354-361)
                    o.x = __fmul_rn(silu_ref(s1.x), s3.x); o.y = __fmul_rn(silu_ref(s1.y), s3.y);
                    o.z = __fmul_rn(silu_ref(s1.z), s3.z); o.w = __fmul_rn(silu_ref(s1.w), s3.w);
                    if (a.y[0]) *(f4*)(a.y[0] + (size_t)col * a.ldy + row) = o;
                    if (a.ys) stream_store_split3(a, col, row, o);
                }
            } else {
                for (uint32_t t = tb + egrp; t < tb + batch && t < nt; t += negrp) {
                    f4 s = tile_sum(t - tb);
                    uint32_t g, tile;
                    tile_of(t0 + t, &g, &tile);
                    const uint32_t row = tile * 16 + quad * 4;
                    if (a.epi == ST_EPI_QKV_ROPE) {   // Rope mode 0 on Q / mode 1 on the new K rows (ml.go:2253-2328), K, V appended (llama.go:274-278)
                        const uint32_t pos = a.rows ? a.rows[col].pos : a.past + col, half = a.hd >> 1;
                        if (g < 2) {
                            const double2 c0 = a.rope[(size_t)pos * half + ((row % a.hd) >> 1)], c1 = a.rope[(size_t)pos * half + (((row + 2) % a.hd) >> 1)];
                            float o0, o1, o2, o3;
                            rope_rotate(s.x, s.y, c0, &o0, &o1);
                            rope_rotate(s.z, s.w, c1, &o2, &o3);
                            s = f4{o0, o1, o2, o3};
                        }
                        float* kcb = a.rows ? a.rows[col].kc + a.kv_off : a.k_cache;
                        float* vcb = a.rows ? a.rows[col].vc + a.kv_off : a.v_cache;
                        float* dst = g == 0 ? a.q_out + (size_t)col * a.M + row : (g == 1 ? kcb : vcb) + (size_t)pos * a.M + row;
                        *(f4*)dst = s;
                    } else {
                        const size_t o = (size_t)col * a.ldy + row;
                        const float* rp = g == 0 ? a.r[0] : (g == 1 ? a.r[1] : a.r[2]);
                        float* yp = (g == 0 ? a.y[0] : (g == 1 ? a.y[1] : a.y[2])) + (size_t)ks * a.ysplit;
                        if (rp) {
                            const f4 rv = *(const f4*)(rp + o);
                            s.x = __fadd_rn(s.x, rv.x); s.y = __fadd_rn(s.y, rv.y); s.z = __fadd_rn(s.z, rv.z); s.w = __fadd_rn(s.w, rv.w);   // Add ml.go:2515-2584
                        }
                        *(f4*)(yp + o) = s;
                        if (a.ys && g == 0) stream_store_split3(a, col, row, s);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// k_stream_mm2 — the same computation with SPECIALISED waves: 8 waves per workgroup, waves 0-3 only move data (global -> registers ->
// LDS image), waves 4-7 only run the matrix cores on the image of the previous chunk; two LDS images, ONE workgroup barrier per chunk.
// Why: in k_stream_mm every wave alternates between issuing a burst of loads and computing.  Its phase trace (tools/stream_mm_check,
// shader clocks per chunk of w1|w3 at 16 rows: lds-barrier 1647 | wait loads 35 | stash+issue 2991 | barrier 244 | compute 1194) shows
// that the loads have always landed by the time they are waited for - the time goes into ISSUING them: two chunks per CU are more than
// the memory pipeline accepts at once, the issue blocks, and with one wave per SIMD a blocked wave also stops feeding its matrix
// core.  Here a loader wave that blocks costs nothing (that is its job) and the MFMA waves never touch global memory.
// Same image layout, operand mapping and summation structure as k_stream_mm (4 compute waves x k-blocks, wave order in the epilogue).
template <int MAXT, int NCT, int KC>
#ifndef STREAM_KERNEL_ATTR
#define STREAM_KERNEL_ATTR           // probe builds: e.g. -DSTREAM_KERNEL_ATTR='__attribute__((amdgpu_waves_per_eu(4,4)))' for two workgroups per CU
#endif
__global__ __launch_bounds__(2 * ST_TH) STREAM_KERNEL_ATTR void k_stream_mm2(const StreamArgs a) {
    LH_TOUCH_ARGS(a.w[0], a.r[2], a.epi, a.gamma, a.ys_plane, a.ldys);   // the argument block's lines behind one wait (kernels_common.h)
    static_assert(KC == 64 || KC == 128 || KC == 256, "chunk");   // 64: four column tiles (49..64 rows) - half-length chunks make room for 6 + 4 tiles twice
    constexpr int ST_PITCH = KC + 4, RPP = 1024 / KC;
    constexpr int NW = MAXT * 16 / RPP, NX = NCT * 16 / RPP;
    constexpr size_t IMG = (size_t)(MAXT + NCT) * 16 * ST_PITCH;      // floats per image
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* img = (float*)smem_raw;                      // [2][IMG]: weights [MAXT * 16][PITCH], then activations [NCT * 16][PITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tiles_per_mat = a.M >> 4, T = tiles_per_mat * a.groups;
    // ST_EPI_SILU_MUL: virtual tile v = (tile v >> 1 of matrix v & 1), dealt in PAIRS, so a workgroup holds w1 and w3 of the same rows
    const bool pairs = a.epi == ST_EPI_SILU_MUL;
    const uint32_t units = pairs ? tiles_per_mat : T, um = pairs ? 2u : 1u;
    const uint32_t S = a.ksplit > 1 ? a.ksplit : 1u, bg = (uint32_t)blockIdx.x / S, ks = (uint32_t)blockIdx.x - bg * S, ng = (uint32_t)gridDim.x / S;
    if (bg >= ng) return;
    const uint32_t t0 = um * (uint32_t)(((uint64_t)bg * units) / ng), t1 = um * (uint32_t)(((uint64_t)(bg + 1) * units) / ng);
    if (t1 <= t0) return;
    auto tile_of = [&](uint32_t v, uint32_t* g, uint32_t* tile) {   // virtual tile -> (matrix, 16-row tile in it)
        if (pairs) { *g = v & 1u; *tile = v >> 1; }
        else { *g = (v >= tiles_per_mat ? 1u : 0u) + (v >= 2 * tiles_per_mat ? 1u : 0u); *tile = v - *g * tiles_per_mat; }   // (<= 3 matrices: no division)
    };
    const uint32_t nt = t1 - t0;
    const uint32_t nch_all = a.K / KC, ch0 = (uint32_t)(((uint64_t)ks * nch_all) / S);
    const uint32_t nch = (uint32_t)(((uint64_t)(ks + 1) * nch_all) / S) - ch0;      // this workgroup's K-chunks (all of them without a split)
    const uint32_t kbase = ch0 * KC;
    if (nch == 0) return;                                                           // (the host keeps S <= K / KC)
    typedef const f4 __attribute__((address_space(1))) gf4;
    constexpr int KB = KC / 64;
    constexpr int KA0 = (MAXT * NCT >= 4) ? 1 : (MAXT * NCT >= 2 ? 2 : 4), KA = KA0 < KB ? KA0 : KB;
    // PIPE (round 3; the half-length chunks of four / five column tiles): the MFMA waves run ONE chunk behind the images - while chunk c
    // multiplies out of registers they read the operands of chunk c + 1 (its image was completed a barrier ago), so no operand read stands
    // between a barrier and the first MFMA behind it.  The images and the loader waves are unchanged: chunk k is stashed before barrier k;
    // chunk k's operands are read between barriers k and k + 1 and multiplied between k + 1 and k + 2, which is also when the loader
    // overwrites its image with chunk k + 2.  Two more barriers per launch.  Measured (profiles/r03_stream_operand_pipeline.txt): with
    // 64-column chunks (a barrier every 96-120 MFMAs) +2..5 % (w2 pairs at 64 rows 75.4 -> 71.2 us, wq|wk|wv 73.7 -> 71.0, w1|w3 129.5 ->
    // 127.4); with 128-column chunks -2.7..+1.5 %, i.e. nothing: the reads behind a barrier are NOT what keeps the matrix pipe at 73 %.
#ifndef STREAM_PIPE
#define STREAM_PIPE 1                // -DSTREAM_PIPE=0: the round-2 schedule (operands read behind each barrier), for A/B builds of the checker
#endif
    constexpr bool PIPE = STREAM_PIPE != 0 && KC == 64 && NCT >= 2 && NCT <= 5;
    f4m acc[KA][MAXT][NCT];
    const uint32_t r16 = (uint32_t)lane & 15, slot = (uint32_t)lane >> 4;
    if (wave < 4) {
        // ---- loader waves
        // (s_setprio 3 here was measured: no effect on any phase, profiles/r02d_stream_traffic_probe.txt - it is the loads' vector address
        // arithmetic that waits for the MFMA burst, see probe 16 below, not the arbitration between the two waves of a SIMD)
        const uint32_t rsub = (uint32_t)tid / (KC / 4), seg = (uint32_t)tid % (KC / 4);
        const float* xp[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            uint32_t c = (uint32_t)i * RPP + rsub;
            c = c < a.n ? c : a.n - 1;
            xp[i] = a.x + (size_t)c * a.ldx + kbase + seg * 4;
        }
#ifndef STREAM_NS
#define STREAM_NS 2
#endif
        // register sets = chunks in flight.  Two everywhere except the single-tile fp32 launches of <= 16 rows (wo, w2: one 16-row tile per CU,
        // 8 / 16 KB per chunk - too few bytes in flight with two): four there (profiles/r03_stream_sets_in_flight.txt: wo 18.6 -> 17.8 us, w2
        // 42.8 -> 38.4; with more tiles three are flat and four much slower: 144+ registers of rows in flight).  -DSTREAM_NS=n overrides for probes.
        constexpr int NS = (STREAM_NS == 2 && MAXT == 1 && NCT == 1) ? 4 : STREAM_NS;
        // gamma chunk: fetched with every set (from x itself when the launch has no norm: the count of loads per set stays a constant)
        const float* gp = (a.gamma ? a.gamma : a.x) + kbase + seg * 4;
        const bool norm = a.gamma != nullptr;
        double ssq[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) ssq[i] = 0.0;
        auto stash_x = [&](const f4 (&xr)[NX], const f4& gq, float* im) {
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                f4 v = xr[i];
                if (norm) {
                    // fp32 squares; four of them meet in fp32 (one more rounding of relative 6e-8 per group), the groups add up in f64:
                    // a quarter of the f64 work of adding every square in f64, which at 32 rows made the loader waves the slower side
                    ssq[i] += (double)__fadd_rn(__fadd_rn(__fmul_rn(v.x, v.x), __fmul_rn(v.y, v.y)), __fadd_rn(__fmul_rn(v.z, v.z), __fmul_rn(v.w, v.w)));
                    v.x = __fmul_rn(gq.x, v.x); v.y = __fmul_rn(gq.y, v.y); v.z = __fmul_rn(gq.z, v.z); v.w = __fmul_rn(gq.w, v.w);
                }
                *(f4*)(im + (size_t)(MAXT * 16 + i * RPP + rsub) * ST_PITCH + seg * 4) = v;
            }
        };
        auto publish_scales = [&]() {    // after the last chunk: per-column sum over the lanes that share a row, then 1 / sqrt(mean + eps)
            if (!norm) return;
            float* scl = img + 2 * IMG;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                double t = ssq[i];
#pragma unroll
                for (int o = KC / 8; o >= 1; o >>= 1) t += __shfl_xor(t, o, 64);
                if (seg == 0) scl[i * RPP + rsub] = (float)(1.0 / sqrt(t / (double)a.K + 1e-5));
            }
        };
        {
        const float* wp[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            uint32_t rr = (uint32_t)i * RPP + rsub;
            rr = rr < nt * 16 ? rr : nt * 16 - 1;
            uint32_t g, tile;
            tile_of(t0 + (rr >> 4), &g, &tile);
            const uint32_t row = tile * 16 + (rr & 15);
            const uint64_t base = (uint64_t)a.w[0] + (g >= 1 ? (uint64_t)a.w[1] - (uint64_t)a.w[0] : 0) + (g == 2 ? (uint64_t)a.w[2] - (uint64_t)a.w[1] : 0);
            wp[i] = (const float*)base + (size_t)row * a.K + kbase + seg * 4;
        }
        // buffer loads: all lanes of register i sit in ONE tile (a pass of the loader covers RPP <= 16 rows and 16 is a multiple
        // of RPP), so the tile's first row is a uniform base; the lane keeps (row in tile) * K + its 16-byte segment as a byte offset
        constexpr bool BUF = NCT >= 2;
        __amdgpu_buffer_rsrc_t wres[NW], xres = stream_rsrc(a.x + kbase), gres = stream_rsrc((a.gamma ? a.gamma : a.x) + kbase);
        uint32_t wvo[NW], xvo[NX];
        if constexpr (BUF) {
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const uint32_t ts0 = ((uint32_t)i * RPP) >> 4, ts = ts0 < nt ? ts0 : nt - 1;   // tile slot of this register: uniform
                uint32_t g, tile;
                tile_of(t0 + ts, &g, &tile);
                const float* mb = g == 0 ? a.w[0] : (g == 1 ? a.w[1] : a.w[2]);
                wres[i] = stream_rsrc(mb + (size_t)tile * 16 * a.K + kbase);
                uint32_t rr = (uint32_t)i * RPP + rsub;
                rr = rr < nt * 16 ? rr : nt * 16 - 1;                                          // same clamp as the pointer path
                wvo[i] = ((rr & 15u) * a.K + seg * 4u) * 4u;
            }
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                uint32_t c = (uint32_t)i * RPP + rsub;
                c = c < a.n ? c : a.n - 1;
                xvo[i] = (c * a.ldx + seg * 4u) * 4u;
            }
        }
        f4 ws[NS][NW], xs[NS][NX], gs[NS];
        // (Round 2-3's timing-only probe builds of this loader - one traffic class taken out at a time, uniform-base addressing - showed that the
        // loads' vector address arithmetic waits for the MFMA burst: profiles/r02d_stream_traffic_probe.txt; they led to the buffer loads below
        // and to k_stream_dma and are no longer part of the kernel.)
        auto issue = [&](f4 (&wr)[NW], f4 (&xr)[NX], f4& gq, uint32_t ch) {
            const uint32_t k0 = (ch < nch ? ch : nch - 1) * KC;
            if constexpr (BUF) {
                const uint32_t so = k0 * 4u;     // bytes, uniform: the soffset operand
#pragma unroll
                for (int i = 0; i < NW; ++i) wr[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wres[i], wvo[i], so, 2 /* nt */));
#pragma unroll
                for (int i = 0; i < NX; ++i) xr[i] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xres, xvo[i], so, 0));
                gq = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(gres, seg * 16u, so, 0));
                return;
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) wr[i] = __builtin_nontemporal_load((gf4*)(uintptr_t)(wp[i] + k0));
#pragma unroll
            for (int i = 0; i < NX; ++i) xr[i] = *(gf4*)(uintptr_t)(xp[i] + k0);
            gq = *(gf4*)(uintptr_t)(gp + k0);
        };
        auto stash = [&](const f4 (&wr)[NW], const f4 (&xr)[NX], const f4& gq, float* im) {
#pragma unroll
            for (int i = 0; i < NW; ++i) *(f4*)(im + (size_t)(i * RPP + rsub) * ST_PITCH + seg * 4) = wr[i];
            stash_x(xr, gq, im);
        };
        constexpr int PER_SET = NW + NX + 1;
        static_assert(PER_SET * (NS - 1) < 64 || STREAM_NS != 2, "vmcnt range");   // (experiment builds with more sets in flight only launch the shapes that fit)
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            issue(ws[q], xs[q], gs[q], (uint32_t)q);
            __builtin_amdgcn_sched_barrier(0);   // keep the issue order (the waits below count on it)
        }
        uint32_t ch = 0;
        for (; ch + NS <= nch; ch += NS) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                wait_vm<(PER_SET * (NS - 1) < 64 ? PER_SET * (NS - 1) : 63)>();
                stash(ws[q], xs[q], gs[q], ((ch + q) & 1) ? img + IMG : img);   // chunk ch + q lives in image (ch + q) & 1
                issue(ws[q], xs[q], gs[q], ch + q + NS);
                __syncthreads();         // barrier `ch + q`: the image holds the chunk; the compute waves are done with what it held before
            }
        }
        const uint32_t rem = nch - ch;   // < NS chunks left, already requested into sets 0..rem-1; nothing new is issued any more
#pragma unroll
        for (int q = 0; q < NS - 1; ++q) {
            if ((uint32_t)q < rem) {
                if (q == 0) wait_vm<(PER_SET * (NS - 1) < 64 ? PER_SET * (NS - 1) : 63)>();
                else if (q == 1) wait_vm<(PER_SET * (NS > 2 ? NS - 2 : 0) < 64 ? PER_SET * (NS > 2 ? NS - 2 : 0) : 63)>();
                else wait_vm<(PER_SET * (NS > 3 ? NS - 3 : 0) < 64 ? PER_SET * (NS > 3 ? NS - 3 : 0) : 63)>();
                stash(ws[q], xs[q], gs[q], ((ch + q) & 1) ? img + IMG : img);
                __syncthreads();
            }
        }
        wait_vm<0>();                    // the clamped tail loads
        publish_scales();
        }
        if constexpr (PIPE) { __syncthreads(); __syncthreads(); }   // the MFMA waves' last two periods (operands of the last chunk, its MFMAs)
    } else {
        // ---- compute waves
        const int cw = wave - 4;
#pragma unroll
        for (int q = 0; q < KA; ++q)
#pragma unroll
            for (int t = 0; t < MAXT; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[q][t][c] = f4m{0.f, 0.f, 0.f, 0.f};
        auto compute = [&](const float* im) {
            const float* Wt = im;
            const float* Xt = im + (size_t)MAXT * 16 * ST_PITCH;
            constexpr int HB = KB >= 2 ? 2 : 1;
#pragma unroll
            for (int h0 = 0; h0 < KB; h0 += HB) {
                f4 bf[HB][NCT], af[HB][MAXT];
#pragma unroll
                for (int hh = 0; hh < HB; ++hh) {
                    const uint32_t koff = (uint32_t)(KB * cw + h0 + hh) * 16 + slot * 4;
#pragma unroll
                    for (int c = 0; c < NCT; ++c) bf[hh][c] = *(const f4*)(Xt + (size_t)(c * 16 + r16) * ST_PITCH + koff);
#pragma unroll
                    for (int t = 0; t < MAXT; ++t) af[hh][t] = *(const f4*)(Wt + (size_t)(t * 16 + r16) * ST_PITCH + koff);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int hh = 0; hh < HB; ++hh)
#pragma unroll
                        for (int t = 0; t < MAXT; ++t)
#pragma unroll
                            for (int c = 0; c < NCT; ++c)
                                acc[(h0 + hh) % KA][t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[hh][t][s], bf[hh][c][s], acc[(h0 + hh) % KA][t][c], 0, 0, 0);
            }
        };
        if constexpr (PIPE) {
            constexpr int NO = MAXT + NCT;
            f4 ops[2][KB][NO];           // [buffer][k-block][A tiles, then B tiles]
            auto read_ops = [&](f4 (&o)[KB][NO], const float* im) {
                const float* Wt = im;
                const float* Xt = im + (size_t)MAXT * 16 * ST_PITCH;
#pragma unroll
                for (int h = 0; h < KB; ++h) {
                    const uint32_t koff = (uint32_t)(KB * cw + h) * 16 + slot * 4;
#pragma unroll
                    for (int c = 0; c < NCT; ++c) o[h][MAXT + c] = *(const f4*)(Xt + (size_t)(c * 16 + r16) * ST_PITCH + koff);
#pragma unroll
                    for (int t = 0; t < MAXT; ++t) o[h][t] = *(const f4*)(Wt + (size_t)(t * 16 + r16) * ST_PITCH + koff);
                }
            };
            auto mfmas = [&](const f4 (&o)[KB][NO]) {
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int h = 0; h < KB; ++h)
#pragma unroll
                        for (int t = 0; t < MAXT; ++t)
#pragma unroll
                            for (int c = 0; c < NCT; ++c)
                                acc[h % KA][t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[h][t][s], o[h][MAXT + c][s], acc[h % KA][t][c], 0, 0, 0);
            };
            __syncthreads();             // barrier 0: image 0 holds chunk 0
            read_ops(ops[0], img);
            __syncthreads();             // barrier 1: image 1 holds chunk 1, image 0 may be overwritten
            for (uint32_t ch = 0; ch < nch; ch += 2) {   // two chunks per trip: the operand buffers swap by name
                // scheduling fences: the reads are ISSUED in front of the MFMAs they hide behind, and no MFMA drifts across the workgroup
                // barrier (pure register work: the compiler otherwise moves the barrier up behind the block's first MFMA, and the wait for the
                // LDS reads that belongs to it with it)
                if (ch + 1 < nch) read_ops(ops[1], img + IMG);
                __builtin_amdgcn_sched_barrier(0);
                mfmas(ops[0]);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();         // barrier ch + 2
                __builtin_amdgcn_sched_barrier(0);
                if (ch + 1 < nch) {
                    if (ch + 2 < nch) read_ops(ops[0], img);
                    __builtin_amdgcn_sched_barrier(0);
                    mfmas(ops[1]);
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();     // barrier ch + 3
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
        for (uint32_t ch = 0; ch < nch; ++ch) {
            __syncthreads();             // barrier `ch`
            compute((ch & 1) ? img + IMG : img);
        }
        }
    }
    __syncthreads();
    // the folded RMSNorm's per-token scales sit behind the image memory (written by the loader waves; read before `part` is touched)
    stream_epilogue<MAXT, NCT>(a, smem_raw, (uint32_t)(2 * IMG), a.gamma ? (const float*)smem_raw + 2 * IMG : nullptr, t0, nt, ks, tiles_per_mat, [&](int t, int c) {
        f4m v = acc[0][t][c];
#pragma unroll
        for (int q = 1; q < KA; ++q) v += acc[q][t][c];
        return v;
    });
}
// ---------------------------------------------------------------------------------------------------------------------------------
// k_stream_dma — the wave-specialised kernel with LDS-DMA loader waves: fp32 weights, two column tiles on (17..96 token rows, where the
// launch is bound inside the CU, not by memory).
// Why (round 3's diagnosis, profiles/r03_stream_mfma_pmc.txt): from 17 rows on k_stream_mm2 takes the SUM of its HBM time and its matrix-pipe
// time per chunk instead of their maximum - the loader wave of a SIMD needs per KB one load, one ds_write_b128 (13 LDS-port cycles each,
// MI355X_MICROARCH "LDS") and the waits between them, and it only gets issue slots in the gaps of the MFMA wave it shares the SIMD with.
// Here a loader wave issues ONE instruction per KB and nothing else: `buffer_load_dwordx4 ... lds` (uniform resource per instruction, the
// lane's byte offset a constant VGPR, the chunk offset an SGPR: no vector ALU, no LDS write, no staging registers), and the chunks in
// flight live in a ring of NIMG LDS images instead of registers.  Measured round 4 (profiles/r04_stream_dma_probe.txt, same box, w1|w3 /
// wq|wk|wv / w2 / wo of 7B): 32 rows 78.5 -> 67.9 / 44.1 -> 33.5 / 52.2 -> 40.0 / 22.2 -> 20.2 us, 48 rows 102.0 -> 87.6 / 57.4 -> 44.7 / 68.2 -> 46.8 / 29.0 -> 21.6,
// 64 rows 127.9 -> 110.3 / 71.2 -> 58.3 / 85.4 -> 59.5 / 34.4 -> 24.6.
//   * image = [(MAXT + NCT) * 16 rows][KC floats], DENSE (an LDS-DMA instruction writes lane i at base + 16 i: no row padding); the operand
//     reads stay conflict-free through a source-side swizzle (the LDS destination is lane-linear, so the permutation goes on the SOURCE
//     address and the same involution on the read, as in k_gemm_glds): 16-byte granule g of image row r is stored at granule position
//     g ^ (r & 15).  A ds_read_b128 lane group (16 lanes: eight rows at slot s, eight at slot s + 1) then touches 16 different positions.
//   * ring: chunk c lives in image c % NIMG.  Loader: wait until ITS DMAs of chunk c have landed (counted vmcnt: the NIMG - 2 younger
//     chunks stay in flight), raw s_barrier c (a __syncthreads would drain the DMAs: an LDS-DMA is a pending LDS write on the VM
//     counter), then request chunk c + NIMG - 1 into the image chunk c - 1 has left.  One workgroup barrier per chunk as in k_stream_mm2.
//   * MFMA waves: k-blocks of a chunk dealt to the four waves, every wave all tiles, partial tiles added in wave order (stream_epilogue):
//     the summation structure of k_stream_mm2.  PIPE: software pipeline over k-blocks - the operands of the next k-block are read while
//     the current one multiplies out of registers (no LDS latency between a barrier and the first MFMA behind it; the loader side is
//     the same either way: chunk c's image is last read in the period behind barrier c).
//   * CS = 2 (97..128 rows = eight column tiles): the MFMA waves are 2 K-groups x 2 column halves - 6 x 4 accumulator tiles per wave
//     instead of 6 x 8, which would not fit; every wave then reads all weight operands of its K-group's k-blocks and half of the
//     activation operands.
// Grouped matrices, (w1, w3) tile pairs, K-split pairs, batched rows and the fused epilogues as in k_stream_mm2; no folded norm (the
// DMA cannot multiply by gamma on the way: the host keeps the <= 16-row launches that fold it on k_stream_mm2).
#ifndef STREAM_DMA_WAUX
#define STREAM_DMA_WAUX 2   // cache policy of the weight DMAs (2 = nt); probe builds override
#endif
__host__ __device__ inline size_t stream_dma_lds_bytes(int maxt, int nct, int kc, int nimg) { return (size_t)nimg * (maxt + nct) * 16 * kc * 4; }

template <int MAXT, int NCT, int KC, int NIMG, bool PIPE, int CS = 1>
__global__ __launch_bounds__(2 * ST_TH) void k_stream_dma(const StreamArgs a) {
    LH_TOUCH_ARGS(a.w[0], a.r[2], a.epi, a.gamma, a.ys_plane, a.ldys);   // the argument block's lines behind one wait (kernels_common.h)
    static_assert(KC == 64 || KC == 128, "chunk");
    static_assert(NIMG >= 2 && NIMG <= 5, "ring");
    constexpr int GR = KC / 4;                  // 16-byte granules per image row
    constexpr int RPI = 64 / GR;                // image rows one DMA instruction covers (1 KB): 4 at KC = 64, 2 at KC = 128
    constexpr int ROWS = (MAXT + NCT) * 16;
    constexpr int NIW = ROWS / RPI / 4;         // DMA instructions per loader wave and chunk ...
    constexpr int JW = MAXT * 16 / RPI / 4;     // ... the first JW of them weight rows, the others activation rows (16 rows = a multiple of 4 RPI)
    static_assert(ROWS % (RPI * 4) == 0, "rows per loader wave");
    constexpr int WAITN = NIW * (NIMG - 2) < 64 ? NIW * (NIMG - 2) : 63;   // (the counter holds 63: a stricter wait is still a correct one)
    constexpr size_t IMGF = (size_t)ROWS * KC;  // floats per image
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* img = (float*)smem_raw;              // [NIMG][ROWS][KC]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tiles_per_mat = a.M >> 4, T = tiles_per_mat * a.groups;
    const bool pairs = a.epi == ST_EPI_SILU_MUL;   // virtual tile v = (tile v >> 1 of matrix v & 1), dealt in PAIRS (k_stream_mm2)
    const uint32_t units = pairs ? tiles_per_mat : T, um = pairs ? 2u : 1u;
    const uint32_t S = a.ksplit > 1 ? a.ksplit : 1u, bg = (uint32_t)blockIdx.x / S, ks = (uint32_t)blockIdx.x - bg * S, ng = (uint32_t)gridDim.x / S;
    if (bg >= ng) return;
    const uint32_t t0 = um * (uint32_t)(((uint64_t)bg * units) / ng), t1 = um * (uint32_t)(((uint64_t)(bg + 1) * units) / ng);
    if (t1 <= t0) return;
    const uint32_t nt = t1 - t0;                // <= MAXT (host)
    const uint32_t nch_all = a.K / KC, ch0 = (uint32_t)(((uint64_t)ks * nch_all) / S);
    const uint32_t nch = (uint32_t)(((uint64_t)(ks + 1) * nch_all) / S) - ch0;      // this workgroup's K-chunks (all of them without a split)
    const uint32_t kbase = ch0 * KC;
    if (nch == 0) return;                                                           // (the host keeps S <= K / KC)
    const uint32_t r16 = (uint32_t)lane & 15, slot = (uint32_t)lane >> 4;
    static_assert(CS == 1 || (CS == 2 && NCT % 2 == 0), "column split");
    constexpr int NKG = 4 / CS, NCW = NCT / CS; // the four MFMA waves = NKG K-groups x CS column parts of NCW tiles (CS = 2: seven / eight column tiles,
                                                // whose MAXT x NCT accumulator tiles would not fit one wave's registers)
    constexpr int KB = KC / 16 / NKG;           // k-blocks (of 16 columns) per MFMA wave and chunk
    f4m acc[MAXT][NCW];
    if (wave < 4) {
        // ---- loader waves: instruction j of wave w covers image rows [(4 j + w) RPI, +RPI): all in ONE 16-row tile, so the tile's first
        // row (at this workgroup's first column) is a uniform resource base; the lane keeps (row in tile) * K + its swizzled granule
        uint32_t voff[NIW];
        const float* base[NIW];
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const uint32_t q = (uint32_t)j * 4 + (uint32_t)wave;
            const uint32_t rr = q * RPI + (uint32_t)lane / GR;          // image row this lane feeds
            const uint32_t gd = (uint32_t)lane % GR;                    // granule position it lands on
            const uint32_t gs = gd ^ (rr & 15u);                        // source granule stored there
            if (j < JW) {
                uint32_t ti = (q * RPI) >> 4;                           // tile slot of the instruction: uniform
                ti = ti < nt ? ti : nt - 1;                             // slots beyond the block: a valid tile again, its sums are never stored
                const uint32_t v = t0 + ti;
                uint32_t g, tile;
                if (pairs) { g = v & 1u; tile = v >> 1; }
                else { g = (v >= tiles_per_mat ? 1u : 0u) + (v >= 2 * tiles_per_mat ? 1u : 0u); tile = v - g * tiles_per_mat; }   // (<= 3 matrices: no division)
                const float* mb = g == 0 ? a.w[0] : (g == 1 ? a.w[1] : a.w[2]);
                base[j] = mb + (size_t)tile * 16 * a.K + kbase;
                voff[j] = ((rr & 15u) * a.K + gs * 4u) * 4u;
            } else {
                uint32_t c = rr - (uint32_t)MAXT * 16;
                c = c < a.n ? c : a.n - 1;                              // columns past the batch: the last row again (never stored)
                base[j] = a.x + kbase;
                voff[j] = (c * a.ldx + gs * 4u) * 4u;
            }
        }
        auto issue = [&](uint32_t ch) {
            const uint32_t cc = ch < nch ? ch : nch - 1;                // past the end: a harmless reload into a free image (uniform counts)
            const uint32_t k0b = cc * (uint32_t)KC * 4u;
            float* im = img + (size_t)(ch % NIMG) * IMGF;
            // (resource and destination as named locals: with the expressions written straight into the builtin's argument list hipcc / ROCm 7.2
            // silently drops the kernel's HOST stub - the library then fails to load with an undefined symbol)
#pragma unroll
            for (int j = 0; j < NIW; ++j) {     // 1 KB per instruction
                const __amdgpu_buffer_rsrc_t rs = stream_rsrc(base[j]);
                __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(im + (size_t)(j * 4 + wave) * 256);
                if (j < JW) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (int)voff[j], (int)k0b, 0, STREAM_DMA_WAUX);   // weights: read once (nt)
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, (int)voff[j], (int)k0b, 0, 0);                        // activations: every workgroup reads them, out of L2
            }
        };
#pragma unroll
        for (int c = 0; c < NIMG - 1; ++c) issue((uint32_t)c);
        for (uint32_t ch = 0; ch < nch; ++ch) {
            wait_vm<WAITN>();                   // chunk ch of this wave has landed; the NIMG - 2 younger ones may still be in flight
            __builtin_amdgcn_s_barrier();       // barrier ch: every part of chunk ch is in its image, and the MFMA waves have left chunk ch - 1's
            issue(ch + NIMG - 1);               // ... whose image takes chunk ch + NIMG - 1
        }
        if constexpr (PIPE) __builtin_amdgcn_s_barrier();   // the MFMA waves' extra period (the last chunk's MFMAs)
        wait_vm<0>();                           // the clamped tail requests
    } else {
        // ---- MFMA waves (k_stream_mm2's structure; operands out of the dense, swizzled image)
        const int cw = wave - 4;
        const uint32_t kq = CS == 2 ? (uint32_t)cw >> 1 : (uint32_t)cw, c0 = CS == 2 ? ((uint32_t)cw & 1u) * NCW : 0u;
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int c = 0; c < NCW; ++c) acc[t][c] = f4m{0.f, 0.f, 0.f, 0.f};
        constexpr int NO = MAXT + NCW;
        auto read_kb = [&](f4 (&o)[NO], const float* im, int h) {   // operands of this wave's k-block h of the chunk in `im`
            const uint32_t g = ((kq * KB + (uint32_t)h) * 4 + slot) ^ r16;   // granule position of (k-block, slot) in a row with r & 15 = r16
#pragma unroll
            for (int c = 0; c < NCW; ++c) o[MAXT + c] = *(const f4*)(im + ((size_t)(MAXT * 16 + (c0 + c) * 16 + r16) * GR + g) * 4);
#pragma unroll
            for (int t = 0; t < MAXT; ++t) o[t] = *(const f4*)(im + ((size_t)(t * 16 + r16) * GR + g) * 4);
        };
        auto mfma_kb = [&](const f4 (&o)[NO]) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < MAXT; ++t)
#pragma unroll
                    for (int c = 0; c < NCW; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(o[t][s], o[MAXT + c][s], acc[t][c], 0, 0, 0);
        };
        auto image = [&](uint32_t ch) { return (const float*)(img + (size_t)(ch % NIMG) * IMGF); };
        if constexpr (PIPE) {
            // software pipeline over k-blocks, two operand sets swapped by name: the operands of the NEXT k-block are requested, then the
            // current one multiplies out of registers.  Scheduling fences: the reads are ISSUED in front of the MFMAs they hide behind and no
            // MFMA drifts across a workgroup barrier.  A chunk's image is last read in the period behind its own barrier.
            static_assert(KB == 1 || KB == 2, "k-blocks per wave and chunk");
            f4 opa[NO], opb[NO];
            barrier_lds_only();                 // barrier 0: chunk 0 is in image 0
            read_kb(opa, img, 0);
            if constexpr (KB == 2) {
                for (uint32_t ch = 0; ch < nch; ++ch) {
                    read_kb(opb, image(ch), 1);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_kb(opa);
                    __builtin_amdgcn_sched_barrier(0);
                    barrier_lds_only();         // barrier ch + 1 (behind the last chunk: the loader's extra one)
                    if (ch + 1 < nch) read_kb(opa, image(ch + 1), 0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_kb(opb);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                for (uint32_t ch = 0; ch < nch; ch += 2) {   // two chunks per trip
                    barrier_lds_only();         // barrier ch + 1
                    if (ch + 1 < nch) read_kb(opb, image(ch + 1), 0);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_kb(opa);
                    __builtin_amdgcn_sched_barrier(0);
                    if (ch + 1 < nch) {
                        barrier_lds_only();     // barrier ch + 2
                        if (ch + 2 < nch) read_kb(opa, image(ch + 2), 0);
                        __builtin_amdgcn_sched_barrier(0);
                        mfma_kb(opb);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else {
            for (uint32_t ch = 0; ch < nch; ++ch) {
                barrier_lds_only();             // barrier ch (this wave's operand reads of chunk ch - 1 are complete: lgkmcnt(0))
                f4 ops[KB][NO];
#pragma unroll
                for (int h = 0; h < KB; ++h) read_kb(ops[h], image(ch), h);
#pragma unroll
                for (int h = 0; h < KB; ++h) mfma_kb(ops[h]);
            }
        }
    }
    __syncthreads();   // the images are dead (the loader waves have drained their DMAs: a pending LDS-DMA would land in `part`)
    stream_epilogue<MAXT, NCT, CS>(a, smem_raw, (uint32_t)(NIMG * IMGF), nullptr, t0, nt, ks, tiles_per_mat, [&](int t, int c) { return acc[t][c]; });
}

// (Block-int8 weights ran on this structure as k_stream_q8 in round 4 - fp32-input MFMAs behind a vector-ALU dequantisation; round 5 replaced
// it by k_stream_q8b, kernels_stream_q8b.h: the bf16 matrix pipe through a lossless split of the activations, 2x faster standalone.)

// Second half of a K-split launch: token row b of  y = resid + ((p_0 + p_1) + ...) + p_{S-1}  (fixed order: bit-reproducible; Add
// ml.go:2515-2584), and - gamma != nullptr - the RMSNorm * weight of that row for the next matmul into h (ml.go:1753-1812, 1877-1914: fp32
// squares, f64 sum, one fp32 scale, two roundings per element - k_rmsnorm_rows' arithmetic on a row that is in registers anyway, so the
// split costs no launch: this kernel stands where the norm's stood).  One workgroup per row, rows of up to 8192 floats in registers.
struct StreamReduceArgs {
    const float* part;   // [S][n][ldy] partial products
    uint64_t stride;     // floats between partials
    const float* resid;  // [n][ldy] or nullptr
    float* y;            // [n][ldy]
    const float* gamma;  // [d] or nullptr: no norm output
    float* h;            // [n][d] normalised rows (or nullptr)
    uint32_t S, d, ldy;
    uint16_t* hs;        // block-int8 on the bf16 pipe (k_stream_q8b): the normalised rows as three bf16 planes, row b at hs + p * hs_plane + b * d
    uint64_t hs_plane;
};
__global__ __launch_bounds__(256) void k_stream_reduce_norm(const StreamReduceArgs a) {
    LH_TOUCH_ARGS(a.part, a.hs);
    __shared__ double sred[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NV = 8;
    const uint32_t d4 = a.d / 4;
    const size_t ro = (size_t)blockIdx.x * a.ldy;
    f4 v[NV], g[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {       // unconditional loads from a clamped index
        const uint32_t i = (uint32_t)tid + (uint32_t)j * 256, ic = i < d4 ? i : 0;
        v[j] = ((const f4*)(a.part + ro))[ic];
        g[j] = a.gamma ? ((const f4*)a.gamma)[ic] : f4{1.f, 1.f, 1.f, 1.f};
    }
    for (uint32_t s = 1; s < a.S; ++s) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const uint32_t i = (uint32_t)tid + (uint32_t)j * 256;
            const f4 q = ((const f4*)(a.part + (size_t)s * a.stride + ro))[i < d4 ? i : 0];
            v[j].x = __fadd_rn(v[j].x, q.x); v[j].y = __fadd_rn(v[j].y, q.y); v[j].z = __fadd_rn(v[j].z, q.z); v[j].w = __fadd_rn(v[j].w, q.w);
        }
    }
    if (a.resid) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const uint32_t i = (uint32_t)tid + (uint32_t)j * 256;
            const f4 q = ((const f4*)(a.resid + ro))[i < d4 ? i : 0];
            v[j].x = __fadd_rn(v[j].x, q.x); v[j].y = __fadd_rn(v[j].y, q.y); v[j].z = __fadd_rn(v[j].z, q.z); v[j].w = __fadd_rn(v[j].w, q.w);
        }
    }
    double ss = 0.0;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const uint32_t i = (uint32_t)tid + (uint32_t)j * 256;
        if (i < d4) {
            ((f4*)(a.y + ro))[i] = v[j];
            ss += (double)__fmul_rn(v[j].x, v[j].x); ss += (double)__fmul_rn(v[j].y, v[j].y);
            ss += (double)__fmul_rn(v[j].z, v[j].z); ss += (double)__fmul_rn(v[j].w, v[j].w);
        }
    }
    if (!a.gamma) return;
    ss = wave_sum_f64(ss);
    if (lane == 0) sred[wave] = ss;
    __syncthreads();
    const double mean = (((sred[0] + sred[1]) + sred[2]) + sred[3]) / (double)a.d;
    const float scale = (float)(1.0 / sqrt(mean + 1e-5));
    float* hr = a.h + (size_t)blockIdx.x * a.d;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const uint32_t i = (uint32_t)tid + (uint32_t)j * 256;
        if (i < d4) {
            f4 o;
            o.x = __fmul_rn(g[j].x, __fmul_rn(v[j].x, scale)); o.y = __fmul_rn(g[j].y, __fmul_rn(v[j].y, scale));
            o.z = __fmul_rn(g[j].z, __fmul_rn(v[j].z, scale)); o.w = __fmul_rn(g[j].w, __fmul_rn(v[j].w, scale));
            if (a.h) ((f4*)hr)[i] = o;
            if (a.hs) {
                uint32_t hh[4], mm[4], ll[4];
                split3(o.x, &hh[0], &mm[0], &ll[0]); split3(o.y, &hh[1], &mm[1], &ll[1]); split3(o.z, &hh[2], &mm[2], &ll[2]); split3(o.w, &hh[3], &mm[3], &ll[3]);
                uint16_t* op = a.hs + (size_t)blockIdx.x * a.d + (size_t)i * 4;
                *(uint2*)(op) = uint2{hh[0] | (hh[1] << 16), hh[2] | (hh[3] << 16)};
                *(uint2*)(op + a.hs_plane) = uint2{mm[0] | (mm[1] << 16), mm[2] | (mm[3] << 16)};
                *(uint2*)(op + 2 * a.hs_plane) = uint2{ll[0] | (ll[1] << 16), ll[2] | (ll[3] << 16)};
            }
        }
    }
}

__host__ __device__ inline size_t stream2_lds_bytes(int maxt, int nct, int kc) { return 2 * stream_lds_bytes(maxt, nct, kc) + 256; }   // + per-column norm scales

}  // namespace lh
