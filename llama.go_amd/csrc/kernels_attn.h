// csrc/kernels_attn.h — single-pass causal attention for prefill (SURVEY App. B `attn_prefill`).
//
// Reference shape: pkg/llama/llama.go:300-333 builds KQ = MulMat(K, Q) for the FULL T x N block, Scale, DiagMaskInf, SoftMax, a
// transposed copy of V, KQV = MulMat(VTrans, KQSoftMax), and a merge copy.  Round 1 mirrored that as four kernels around a
// materialised score tensor S[H][N][T] and a V^T copy: 13B, N = 1024 spent 315 us per layer there and moved 168 MB of scores.
// Here one kernel walks the keys once per query block with an online softmax: no S tensor, no V^T copy, no second pass.
//
// Mapping onto v_mfma_f32_32x32x2_f32 (exact f32 fma chains, like every matrix-core path of this backend):
//   S^T[key][query] = K[key][c] . Q[query][c]      A = K tile rows (from LDS), B = Q fragment (registers, loaded once per block)
//   O^T[c][query]  += V^T[c][key] . P^T[key][query] A = V tile (from LDS),      B = P^T — which IS the accumulator layout of S^T:
// lane l owns query column l % 32 in both products, so the probabilities never move between lanes, the running max / sum / rescale
// are per-lane scalars, and the only cross-lane traffic of the softmax is one exchange between the two half-waves per tile.
//
// Work item = (head, block of 64 queries); a workgroup has two wave PAIRS: the pair splits the 64 queries (interleaved, so both
// waves see the same causal extent), and the two pairs take the even / odd 32-key tiles with their own (m, l, O) state, merged
// once per item through LDS.  K and V tiles travel global -> LDS by LDS-DMA (no staging registers); K rows are XOR-swizzled on the
// source side so that 32 lanes reading the same 16-byte granule of 32 different rows hit 32 different slots.
//
// Balance: a 64-query block costs ceil(visible keys / 64) steps, so the causal triangle makes the last blocks the longest (13B, N = 1024:
// 640 blocks of 1..16 steps on 512 workgroups, mean 10.6) and a chunk of a long conversation has few, very long ones (64 queries behind
// 1984 cached keys: 32 blocks of 32 steps).  The host (plan.hip, attention_flash) therefore cuts the blocks above a chosen length into
// PARTS by key range, orders all parts longest first (FlashArgs::work) and the workgroups draw them back and forth (round 0 left to
// right, round 1 right to left, ...).  An uncut block is finished here as before; a part leaves its unnormalised (O, m, l) per query in
// FlashArgs::part and k_attn_flash_combine adds the parts of a block in part order - a fixed partition and a fixed order, so results
// do not depend on which workgroup ran what.
// The K tiles of step s+1 are requested while step s multiplies P.V, the V tiles while step s+1 multiplies K.Q.
//
// Rounding: s = fl32(dot) * fl32(1/sqrt(hd)) as in ml.go:2371; p = exp(fl32(s - m)) with m the RUNNING maximum, evaluated by the fp32
// library exponential (<= 1 ulp; the reference rounds a float64 exp, ml.go:2472-2492 — sixteen float64 exponentials per lane and tile
// cost half of the matrix-pipe time and 120 registers here), and the normalisation by the fp32 sum happens once at the end (the
// reference normalises p before the P.V product and sums in key order): the same value up to a few fp32 roundings per output
// (~1e-6 relative against the checker, tolerance 1e-4).  Masked keys contribute exactly 0 (ml.go:2476-2477).
#pragma once
#include "kernels_llama.h"
#include "attn_worklist.h"

namespace lh {

typedef float f16acc __attribute__((ext_vector_type(16)));

struct FlashArgs {
    const float* q;        // [n][d] roped queries (row stride d)
    const float* k_cache;  // this layer's K slot [ctx][d] (post-RoPE keys)
    const float* v_cache;
    float* out;            // [n][d] merged heads
    uint32_t d, H, n, past;
    float scale;           // fl32(1/sqrt(hd)) llama.go:306
    uint32_t nqb;          // query blocks of FA_BQ
    // work decomposition (see "Balance" above)
    float* part;           // [H][nqb - qb_cut][pmax][FA_BQ][FA_PSTRIDE] partial results of the blocks that are cut; null when none is
    uint32_t chunk;        // a block of more than `chunk` steps is cut into ceil(steps / chunk) parts of near-equal length; 0 = no block is cut
    uint32_t qb_cut;       // first block that is cut (the step count grows with the block index)
    uint32_t pmax;         // parts of the longest block
    uint32_t nwork;        // entries of work[]; 0 = no list: blocks in descending order, uncut
    uint16_t work[FA_MAXW];   // (block << 4 | part), longest first; every head runs the same list (attn_worklist.h)
};

constexpr int FA_TH = 256, FA_HD = 128;
constexpr int FA_PSTRIDE = FA_HD + 4;   // a partial record per query: 128 O values, m, l, padding to 16 bytes
constexpr size_t FA_LDS_BYTES = (size_t)4 * 32 * FA_HD * 4;   // K tiles of the two pairs + V tiles of the two pairs = 64 KiB

// (Tried and dropped, round 3: s_nop 2 / 8 / 16 behind every MFMA so that the other workgroup's softmax issues next to them - 0.792 / 0.792 /
// 0.791 of the fp32 peak on 13B N = 1024 against 0.793 unpaced, profiles/r03_attn_pace.txt.)
__global__ __launch_bounds__(FA_TH) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_attn_flash(const FlashArgs a) {
    extern __shared__ __attribute__((aligned(16))) float fa_smem[];
    float* Ksm = fa_smem;                     // [2][32][128], granule g of row r stored at slot g ^ r
    float* Vsm = fa_smem + 2 * 32 * FA_HD;    // [2][32][128], plain
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = wave >> 1, qhalf = wave & 1;
    const int lj = lane & 31, lh2 = lane >> 5;
    const uint32_t d = a.d, T = a.past + a.n;
    const uint32_t items = (a.nwork ? a.nwork : a.nqb) * a.H;
    // Items (parts of blocks, every head) are numbered longest first and dealt to the workgroups in boustrophedon order: the workgroups
    // that drew the longest items of one round get the shortest of the next (13B, N = 1024, uncut: 640 items on 512 workgroups; dealt
    // round-robin the 16-step items shared a workgroup with 4-step ones, 20 steps on the longest chain against 16).
    for (uint32_t round = 0;; ++round) {
        const uint32_t item = round * gridDim.x + ((round & 1u) ? gridDim.x - 1u - blockIdx.x : blockIdx.x);
        if (item >= items) break;   // (every later round lies beyond this index too)
        const uint32_t ent = item / a.H, h = item % a.H;
        const uint32_t code = a.nwork ? (uint32_t)a.work[ent] : (a.nqb - 1 - ent) << 4;
        const uint32_t qb = code >> 4, part = code & 15u;
        const uint32_t q0 = qb * FA_BQ;
        const uint32_t qend = q0 + FA_BQ < a.n ? q0 + FA_BQ : a.n;
        const uint32_t Tb = a.past + qend;                 // keys any query of this block can see: 0 .. Tb - 1
        const uint32_t NT = (Tb + 31) / 32, nsteps = (NT + 1) / 2;   // = fa_steps(past, n, qb)
        const uint32_t nparts = a.nwork ? fa_parts(nsteps, a.chunk) : 1u;
        const uint32_t st0 = fa_part_begin(nsteps, nparts, part), st1 = fa_part_begin(nsteps, nparts, part + 1);   // this part's steps
        const uint32_t qi = q0 + 2 * (uint32_t)lj + (uint32_t)qhalf;   // this lane's query (both half-waves hold it)
        const bool qok = qi < a.n;
        const uint32_t qlim = a.past + qi;                 // last visible key of the query
        const float* Kh = a.k_cache + (size_t)h * FA_HD;
        const float* Vh = a.v_cache + (size_t)h * FA_HD;

        // ---- LDS-DMA: per step 4 tile images (K even, K odd, V even, V odd) of 16 KiB; a wave-instruction moves 1 KiB = 2 tile rows;
        // each wave moves 8 pieces of K and 8 of V per step: piece pc of wave w covers image (pc / 4 of its operand), rows 8 (pc % 4)...
        auto dma = [&](bool isK, uint32_t st) {
#pragma unroll
            for (int pc = 0; pc < 8; ++pc) {
                const int img = pc >> 2;                                          // pair whose tile this is
                const uint32_t row = (uint32_t)((pc & 3) * 8 + wave * 2 + lh2);  // tile row written by this half-wave (0..31)
                uint32_t key = (2 * st + (uint32_t)img) * 32 + row;
                key = key < T ? key : T - 1;                                      // beyond the cache extent: a valid row, masked later
                const uint32_t gran = isK ? ((uint32_t)lj ^ row) : (uint32_t)lj;  // source-side swizzle for K
                const float* src = (isK ? Kh : Vh) + (size_t)key * d + 4 * gran;
                float* dst = (isK ? Ksm : Vsm) + img * (32 * FA_HD) + ((pc & 3) * 8 + wave * 2) * FA_HD;   // wave-uniform base, lanes write linearly
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        };
        __builtin_amdgcn_s_barrier();   // everybody is done with the previous item's LDS (tiles and merge area)
        dma(true, st0);
        dma(false, st0);

        // ---- Q fragment: granule 2g + h of the query row, g = 0..15 (B operand of the 4 MFMAs of granule g)
        f4 qf[16];
        {
            const float* qp = a.q + (size_t)(qok ? qi : a.n - 1) * d + (size_t)h * FA_HD + 4 * lh2;
#pragma unroll
            for (int g = 0; g < 16; ++g) qf[g] = *(const f4*)(qp + 8 * g);
        }
        f16acc o[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[ct][e] = 0.f;
        float m = -INFINITY, l = 0.f;

        for (uint32_t st = st0; st < st1; ++st) {
            const uint32_t kt = 2 * st + (uint32_t)pair;       // this pair's key tile
            const bool live = kt < NT;                         // (an odd tile count leaves pair 1 idle in the last step)
            // K(st) landed: my 8 newer V pieces may still be in flight
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            f16acc s;
#pragma unroll
            for (int e = 0; e < 16; ++e) s[e] = 0.f;
            if (live) {
                const float* Kt = Ksm + pair * (32 * FA_HD) + lj * FA_HD;
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const f4 kf = *(const f4*)(Kt + (((2 * g + lh2) ^ lj) << 2));
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.x, qf[g].x, s, 0, 0, 0); 
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.y, qf[g].y, s, 0, 0, 0); 
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.z, qf[g].z, s, 0, 0, 0); 
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf.w, qf[g].w, s, 0, 0, 0); 
                }
            }
            // ---- online softmax on this lane's 16 keys of the tile: key of accumulator entry e = 32 kt + 8 (e / 4) + 4 h + e % 4
            float p[16];
            float mx = -INFINITY;
            const uint32_t kbase = kt * 32 + 4 * (uint32_t)lh2;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const uint32_t key = kbase + 8 * (e >> 2) + (e & 3);
                const bool vis = live && qok && key <= qlim;   // DiagMaskInf: key > past + query is masked (ml.go:2401-2404)
                p[e] = vis ? __fmul_rn(s[e], a.scale) : -INFINITY;   // Scale ml.go:2331-2374
                mx = fmaxf(mx, p[e]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));            // the other half-wave holds the other 16 keys of this query
            const float mn = fmaxf(m, mx);
            float alpha = 1.f, psum = 0.f;
            if (mn != -INFINITY) {                             // at least one visible key so far
                alpha = m == -INFINITY ? 0.f : expf(__fsub_rn(m, mn));
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    p[e] = p[e] == -INFINITY ? 0.f : expf(__fsub_rn(p[e], mn));   // ml.go:2472-2492
                    psum += p[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) p[e] = 0.f;
            }
            l = fmaf(l, alpha, psum);
            m = mn;
            if (__any(alpha != 1.f)) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[ct][e] *= alpha;
            }
            // V(st) landed, and every wave is done reading K(st): the K tiles of the next step may come in under P.V
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            dma(true, st + 1 < st1 ? st + 1 : st);             // past the end: a harmless reload (keeps the counts uniform)
            if (live) {
                const float* Vt = Vsm + pair * (32 * FA_HD) + 4 * lj;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int rr = 8 * (e >> 2) + 4 * lh2 + (e & 3);   // tile row of the key this half-wave supplies in step e
                    const f4 vf = *(const f4*)(Vt + rr * FA_HD);
                    o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.x, p[e], o[0], 0, 0, 0); 
                    o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.y, p[e], o[1], 0, 0, 0); 
                    o[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.z, p[e], o[2], 0, 0, 0); 
                    o[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf.w, p[e], o[3], 0, 0, 0); 
                }
            }
            __builtin_amdgcn_s_barrier();                       // every wave is done reading V(st)
            dma(false, st + 1 < st1 ? st + 1 : st);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the redundant tail DMA
        __builtin_amdgcn_s_barrier();

        // ---- merge the two pairs (even / odd key tiles) through LDS: pair 1 publishes (m, l, O), pair 0 combines and stores
        l += __shfl_xor(l, 32, 64);                             // both half-waves now hold the query's whole partial sum
        float* mo = fa_smem + qhalf * (64 * 66);                // [lane][66]: 64 O values + m + l   (2 x 16.5 KiB, the tiles are dead)
        if (pair == 1) {
            float* dst = mo + lane * 66;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[ct * 16 + e] = o[ct][e];
            dst[64] = m;
            dst[65] = l;
        }
        __syncthreads();
        if (pair == 0 && qok) {
            const float* src = mo + lane * 66;
            const float m1 = src[64], l1 = src[65];
            const float M = fmaxf(m, m1);                       // an uncut block: finite (key 0 is visible to every query and lives in an even tile)
            const float a0 = m == -INFINITY ? 0.f : expf(__fsub_rn(m, M));
            const float a1 = m1 == -INFINITY ? 0.f : expf(__fsub_rn(m1, M));
            const float lsum = fmaf(l1, a1, __fmul_rn(l, a0));
            // an uncut block: normalise and store (ml.go:2496-2499: p *= 1/sum).  A part: the unnormalised record for k_attn_flash_combine
            // (a part none of whose keys this query sees leaves O = 0, l = 0, m = -inf).
            const bool whole = nparts == 1;
            const float inv = whole ? __fdiv_rn(1.0f, lsum) : 1.0f;
            float* orow = whole ? a.out + (size_t)qi * d + (size_t)h * FA_HD
                                : a.part + ((((size_t)h * (a.nqb - a.qb_cut) + (qb - a.qb_cut)) * a.pmax + part) * FA_BQ + (qi - q0)) * FA_PSTRIDE;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c4 = 4 * (8 * (e >> 2) + 4 * lh2 + (e & 3));   // accumulator entry e of O^T tile ct = output column c4 + ct
                f4 r;
                r.x = __fmul_rn(fmaf(src[0 * 16 + e], a1, __fmul_rn(o[0][e], a0)), inv);
                r.y = __fmul_rn(fmaf(src[1 * 16 + e], a1, __fmul_rn(o[1][e], a0)), inv);
                r.z = __fmul_rn(fmaf(src[2 * 16 + e], a1, __fmul_rn(o[2][e], a0)), inv);
                r.w = __fmul_rn(fmaf(src[3 * 16 + e], a1, __fmul_rn(o[3][e], a0)), inv);
                *(f4*)(orow + c4) = r;
            }
            if (!whole && lh2 == 0) { orow[FA_HD] = M; orow[FA_HD + 1] = lsum; }
        }
    }
}

// The parts of a cut block, added in part order: out = sum_p O_p e^(m_p - M) / sum_p l_p e^(m_p - M), M = max_p m_p (finite: part 0 holds
// key 0).  grid (H, nqb - qb_cut), 256 threads = 64 queries x 4 column groups of 8 float4.
__global__ __launch_bounds__(256) void k_attn_flash_combine(const FlashArgs a) {
    const uint32_t h = blockIdx.x, qb = a.qb_cut + blockIdx.y;
    const uint32_t nparts = fa_parts(fa_steps(a.past, a.n, qb), a.chunk);
    const uint32_t ql = threadIdx.x >> 2, cg = threadIdx.x & 3u, qi = qb * FA_BQ + ql;
    if (nparts == 1 || qi >= a.n) return;
    const float* rec = a.part + ((((size_t)h * (a.nqb - a.qb_cut) + (qb - a.qb_cut)) * a.pmax) * FA_BQ + ql) * FA_PSTRIDE;   // part p: + p * FA_BQ * FA_PSTRIDE
    float M = -INFINITY;
    for (uint32_t p = 0; p < nparts; ++p) M = fmaxf(M, rec[(size_t)p * FA_BQ * FA_PSTRIDE + FA_HD]);
    f4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    for (uint32_t p = 0; p < nparts; ++p) {
        const float* r = rec + (size_t)p * FA_BQ * FA_PSTRIDE;
        const float mp = r[FA_HD];
        const float w = mp == -INFINITY ? 0.f : expf(__fsub_rn(mp, M));
        den = fmaf(r[FA_HD + 1], w, den);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f4 v = *(const f4*)(r + 4 * (4 * k + (int)cg));
            acc[k].x = fmaf(v.x, w, acc[k].x);
            acc[k].y = fmaf(v.y, w, acc[k].y);
            acc[k].z = fmaf(v.z, w, acc[k].z);
            acc[k].w = fmaf(v.w, w, acc[k].w);
        }
    }
    const float inv = __fdiv_rn(1.0f, den);
    float* orow = a.out + (size_t)qi * a.d + (size_t)h * FA_HD;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        f4 v;
        v.x = __fmul_rn(acc[k].x, inv); v.y = __fmul_rn(acc[k].y, inv); v.z = __fmul_rn(acc[k].z, inv); v.w = __fmul_rn(acc[k].w, inv);
        *(f4*)(orow + 4 * (4 * k + (int)cg)) = v;
    }
}

}  // namespace lh
