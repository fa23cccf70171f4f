// tools/kernels_stream_mm_r1.h — k_stream_mm, round 1's weight-streaming MFMA kernel in which every wave loads AND multiplies (the structure
// k_stream_mm2 / k_stream_dma replaced): removed from the product in round 6 (no LLaMA shape reached it: tests/test_gpu_zz_routes.py);
// tools/stream_mm_check.hip still times it.  Include behind csrc/kernels_stream.h.
#pragma once
namespace lh {
template <int MAXT, int NCT, int KC>
__global__ __launch_bounds__(ST_TH) void k_stream_mm(const StreamArgs a) {
    static_assert(KC == 128 || KC == 256 || KC == 512, "chunk");
    constexpr int ST_KC = KC, ST_PITCH = KC + 4;
    constexpr int RPP = 1024 / KC;                      // image rows one pass of the 256 threads covers (a wave: 64 x 16 B of ONE row from KC = 256)
    constexpr int NW = MAXT * 16 / RPP, NX = NCT * 16 / RPP;   // float4 per thread and chunk: weights, activations
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Wt = (float*)smem_raw;                       // [MAXT * 16][ST_PITCH]
    float* Xt = Wt + (size_t)MAXT * 16 * ST_PITCH;      // [NCT * 16][ST_PITCH]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t tiles_per_mat = a.M >> 4, T = tiles_per_mat * a.groups;
    const uint32_t t0 = (uint32_t)(((uint64_t)blockIdx.x * T) / gridDim.x), t1 = (uint32_t)(((uint64_t)(blockIdx.x + 1) * T) / gridDim.x);
    if (t1 <= t0) return;
    const uint32_t nt = t1 - t0;                        // <= MAXT (host)
    const uint32_t nch = a.K / ST_KC;
    typedef const f4 __attribute__((address_space(1))) gf4;

    // ---- this thread's share of a chunk: rows i*8 + (tid >> 5) of the weight image (and columns of the activation image), float4 tid & 31
    const uint32_t rsub = (uint32_t)tid / (KC / 4), seg = (uint32_t)tid % (KC / 4);
    const float* wp[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t rr = (uint32_t)i * RPP + rsub;
        rr = rr < nt * 16 ? rr : nt * 16 - 1;            // rows past this workgroup's tiles: a duplicate load, never used
        const uint32_t v = t0 * 16 + rr, g = v / a.M, row = v - g * a.M;
        const uint64_t base = (uint64_t)a.w[0] + (g >= 1 ? (uint64_t)a.w[1] - (uint64_t)a.w[0] : 0) + (g == 2 ? (uint64_t)a.w[2] - (uint64_t)a.w[1] : 0);
        wp[i] = (const float*)base + (a.tiled ? (size_t)row * KC : (size_t)row * a.K) + seg * 4;
    }
    const size_t wstep = a.tiled ? (size_t)a.M * KC : (size_t)KC;   // floats between consecutive chunks of a row
    const float* xp[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        uint32_t c = (uint32_t)i * RPP + rsub;
        c = c < a.n ? c : a.n - 1;
        xp[i] = a.x + (size_t)c * a.ldx + seg * 4;
    }
    f4 wa[NW], xa[NX], wb[NW], xb[NX];
    auto issue = [&](f4 (&wr)[NW], f4 (&xr)[NX], uint32_t ch) {
        const uint32_t cc = ch < nch ? ch : nch - 1;                // past the end: the last chunk again (never stored)
        const uint32_t k0 = cc * ST_KC;
        const size_t w0 = (size_t)cc * wstep;
#pragma unroll
        for (int i = 0; i < NW; ++i) wr[i] = __builtin_nontemporal_load((gf4*)(uintptr_t)(wp[i] + w0));
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[i] = *(gf4*)(uintptr_t)(xp[i] + k0);
    };
    auto stash = [&](const f4 (&wr)[NW], const f4 (&xr)[NX]) {
#pragma unroll
        for (int i = 0; i < NW; ++i) *(f4*)(Wt + (size_t)(i * RPP + rsub) * ST_PITCH + seg * 4) = wr[i];
#pragma unroll
        for (int i = 0; i < NX; ++i) *(f4*)(Xt + (size_t)(i * RPP + rsub) * ST_PITCH + seg * 4) = xr[i];
    };
    // KA independent accumulator sets per (tile, column tile), k-blocks dealt to them in turn: with one or two tiles a single
    // accumulator makes every MFMA wait for its predecessor (40 cycles dependent vs 32 issue) and the chunk's matrix work a serial chain
    constexpr int KB = KC / 64;                         // k-blocks of a chunk per wave
    constexpr int KA0 = (MAXT * NCT >= 4) ? 1 : (MAXT * NCT >= 2 ? 2 : 4), KA = KA0 < KB ? KA0 : KB;
    f4m acc[KA][MAXT][NCT];
#pragma unroll
    for (int q = 0; q < KA; ++q)
#pragma unroll
        for (int t = 0; t < MAXT; ++t)
#pragma unroll
            for (int c = 0; c < NCT; ++c) acc[q][t][c] = f4m{0.f, 0.f, 0.f, 0.f};
    const uint32_t r16 = (uint32_t)lane & 15, slot = (uint32_t)lane >> 4;
    auto compute = [&]() {
        // Straight-line: all MAXT tiles, also the ones past this workgroup's count (their image rows hold a duplicate of the last
        // row and their sums are dropped).  A branch per tile kept the operand reads next to their MFMAs (LDS latency exposed)
        // and made the compiler drain ALL loads in flight at the loop head; the matrix pipe has the slack (<= 65 % busy).
        // Operands of a pair of k-blocks are read together, ahead of their MFMAs.
        constexpr int HB = KB >= 2 ? 2 : 1;
#pragma unroll
        for (int h0 = 0; h0 < KB; h0 += HB) {
            f4 bf[HB][NCT], af[HB][MAXT];
#pragma unroll
            for (int hh = 0; hh < HB; ++hh) {
                const uint32_t koff = (uint32_t)(KB * wave + h0 + hh) * 16 + slot * 4;
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
    // ---- main stream: two chunks in flight in registers, one in LDS under the matrix cores
    // (the chunk count is even - host check - so the loop body is the same straight line every time and the compiler can count the
    // loads in flight: the wait in front of a stash leaves the OTHER register set's chunk in flight)
    constexpr int PER_SET = NW + NX;
    static_assert(PER_SET < 64, "vmcnt range");
    issue(wa, xa, 0);
    __builtin_amdgcn_sched_barrier(0);   // keep the issue order: the scheduler swapped the two groups, and the first stash then had to drain both
    issue(wb, xb, 1);
    __builtin_amdgcn_sched_barrier(0);
    for (uint32_t ch = 0; ch + 1 < nch; ch += 2) {
        barrier_lds_only();              // everybody is done with the image of the previous chunk
        wait_vm<PER_SET>();
        stash(wa, xa);
        issue(wa, xa, ch + 2);
        __syncthreads();
        compute();
        barrier_lds_only();
        wait_vm<PER_SET>();
        stash(wb, xb);
        issue(wb, xb, ch + 3);
        __syncthreads();
        compute();
    }
    if (nch & 1) {                       // odd chunk count: the last chunk sits in the first register set
        barrier_lds_only();
        wait_vm<PER_SET>();
        stash(wa, xa);
        __syncthreads();
        compute();
    }
    __syncthreads();
    // ---- the four waves' partial tiles meet in LDS: part[tile in batch][wave][column][16 rows]; thread (column, row quad) adds them in wave order
    constexpr int NC = NCT * 16;
    float* part = (float*)smem_raw;
    constexpr uint32_t TILE_FLOATS = 4u * NC * 16;
    const uint32_t batch = (uint32_t)(stream_lds_bytes(MAXT, NCT, KC) / (TILE_FLOATS * 4));   // >= 1: (MAXT + NCT) * 16 * 132 >= 64 * NCT * 16
    for (uint32_t tb = 0; tb < nt; tb += batch) {
#pragma unroll
        for (int t = 0; t < MAXT; ++t) {
            if ((uint32_t)t >= tb && (uint32_t)t < tb + batch && (uint32_t)t < nt) {
#pragma unroll
                for (int c = 0; c < NCT; ++c) {
                    f4m v = acc[0][t][c];
#pragma unroll
                    for (int q = 1; q < KA; ++q) v += acc[q][t][c];
                    *(f4m*)(part + (size_t)(t - tb) * TILE_FLOATS + ((size_t)wave * NC + c * 16 + r16) * 16 + slot * 4) = v;
                }
            }
        }
        __syncthreads();
        const uint32_t col = (uint32_t)tid >> 2, quad = (uint32_t)tid & 3;
        if (col < (uint32_t)NC && col < a.n) {
            for (uint32_t t = tb; t < tb + batch && t < nt; ++t) {
                const float* p = part + (size_t)(t - tb) * TILE_FLOATS + (size_t)col * 16 + quad * 4;
                f4 s = *(const f4*)p;
#pragma unroll
                for (int w = 1; w < 4; ++w) {
                    const f4 q = *(const f4*)(p + (size_t)w * NC * 16);
                    s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
                }
                const uint32_t v = (t0 + t) * 16 + quad * 4, g = v / a.M, row = v - g * a.M;
                const size_t o = (size_t)col * a.ldy + row;
                const float* rp = g == 0 ? a.r[0] : (g == 1 ? a.r[1] : a.r[2]);
                float* yp = g == 0 ? a.y[0] : (g == 1 ? a.y[1] : a.y[2]);
                if (rp) {
                    const f4 rv = *(const f4*)(rp + o);
                    s.x = __fadd_rn(s.x, rv.x); s.y = __fadd_rn(s.y, rv.y); s.z = __fadd_rn(s.z, rv.z); s.w = __fadd_rn(s.w, rv.w);   // Add ml.go:2515-2584
                }
                *(f4*)(yp + o) = s;
            }
        }
        __syncthreads();
    }
}

}  // namespace lh
