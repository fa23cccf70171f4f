// MOVED OUT OF THE PRODUCT in round 3 (was csrc/): the resident-kernel experiment measured a tie with the per-layer kernels and produced one
// unexplained wrong token (profiles/r02b_resident_trace.txt, r02c_resident_trace_again.txt); kept only for tools/persist_probe.hip / resident_probe.hip.
// csrc/kernels_persist.h — building blocks of the PERSISTENT decode kernel (gfx950): one workgroup per CU stays resident for a
// whole token and walks the GEMV phases of llama.Eval (pkg/llama/llama.go:246-384) with a grid barrier between them instead of a
// kernel boundary.
//
// Why: a decode GEMV kernel of 12-57 us pays ~3 us of ramp + tail at every launch boundary (profiles/r02_f32_kernel_trace.txt:
// t = 3 us + bytes / 7.0 TB/s, 129 launches per token = 9 % of a token; for block-int8 the streams are 3.5x shorter and the same
// 3 us is a quarter of each kernel).  The weights of the NEXT phase do not depend on anything computed in this one, so a resident
// workgroup can request its first rows of the next matrix BEFORE it waits at the barrier: they travel global -> LDS by LDS-DMA
// (no registers held across the barrier) and the HBM pipe stays busy while the 16 KB activation vector is handed over.
//
//   phase p:   x -> registers | rows parked in LDS by the previous phase | ring-streamed rows | epilogue, y -> HBM
//   boundary:  stores complete (vmcnt) -> one atomic arrive per workgroup -> LDS-DMA of phase p+1's first rows
//              -> wave 0 polls the arrival counter (scalar path: not queued behind the DMA in the vector-memory FIFO) -> s_barrier
//
// Cross-XCD visibility of the exchanged vectors (each XCD has its own L2): selected by XM
//   XM_FENCE  plain accesses + agent-scope release (buffer_wbl2 sc1) / acquire (buffer_inv sc1) around the barrier
//   XM_SCOPED every exchanged word is read / written with agent-scope (sc1) accesses, no cache maintenance
//   XM_PLAIN  plain accesses, no cache maintenance: for exchange buffers in uncached (MTYPE_UC) or fine-grained memory
#pragma once
#include "../llama.go_amd/csrc/kernels_llama.h"

namespace lh {

enum { XM_FENCE = 0, XM_SCOPED = 1, XM_PLAIN = 2 };
enum { POLL_VECTOR = 0, POLL_SCALAR = 1 };   // arrival counter read by a vector atomic load / through the scalar cache path (s_load glc)

constexpr int PTH = 512;            // threads of the persistent workgroup (8 waves, 2 per SIMD: 256 VGPRs each)
constexpr int PNW = PTH / 64;
constexpr uint32_t P_RED_OFF = 64;          // LDS: [0,64) f64 wave partials of the RMSNorm, then per-row wave partials [rows][PNW]
constexpr uint32_t P_LDS_BYTES = 96 * 1024; // > 80 KiB: one workgroup per CU
constexpr uint32_t P_MAX_ROWS = 1024;       // rows one workgroup may own in a phase (32 KB of partials)

struct PersistCtl {
    unsigned long long* count;   // arrival counter (monotonic; every launch adds exactly arrivals_per_launch)
    uint32_t* err;               // != 0: a barrier timed out; the results are garbage
    unsigned long long arrivals_per_launch;
    uint32_t timeout_ticks;      // per barrier, in 100 MHz ticks
    uint32_t nowait;             // probe only: arrive but never wait (results are garbage; shows the stream's ceiling)
    const float* dummy;          // cache-resident, >= max K floats: target of the ring's out-of-range refills
#ifdef PERSIST_TRACE
    unsigned long long* trace;   // tools/resident_probe: [3][PERSIST_TRACE_MAX] realtime stamps of workgroups 0, #wg/2, #wg-1
#endif
};
#ifdef PERSIST_TRACE
constexpr uint32_t PERSIST_TRACE_MAX = 4096;
#define PERSIST_STAMP(c, wg, nwg, idx) do { if (threadIdx.x == 0 && ((wg) == 0 || (wg) == (nwg) / 2 || (wg) == (nwg) - 1)) { \
        const uint32_t slot_ = (wg) == 0 ? 0u : (wg) == (nwg) - 1 ? 2u : 1u; \
        if ((idx) < PERSIST_TRACE_MAX) (c).trace[slot_ * PERSIST_TRACE_MAX + (idx)] = __builtin_amdgcn_s_memrealtime(); } ++(idx); } while (0)
#else
#define PERSIST_STAMP(c, wg, nwg, idx) do { } while (0)
#endif

// Pointers read out of descriptor tables in memory are generic to the compiler (flat_load: both counters, no scalar base);
// everything this kernel touches outside LDS is global memory, so say so.
typedef const __attribute__((address_space(1))) f4* gptr_f4;
typedef const __attribute__((address_space(1))) float* gptr_f;
typedef __attribute__((address_space(1))) float* gptr_fw;
__device__ __forceinline__ f4 ld_nt_g(const char* p) { return __builtin_nontemporal_load((gptr_f4)p); }

// ---- exchange-vector accesses -------------------------------------------------------------------------------------------------
template <int XM>
__device__ __forceinline__ float ldx1(const float* p) {
    if (XM == XM_SCOPED) return __hip_atomic_load((gptr_f)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *(gptr_f)p;
}
template <int XM>
__device__ __forceinline__ f4 ldx4(const float* p) {
    if (XM == XM_SCOPED) {
        f4 v;
        v.x = __hip_atomic_load((gptr_f)p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.y = __hip_atomic_load((gptr_f)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.z = __hip_atomic_load((gptr_f)p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.w = __hip_atomic_load((gptr_f)p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    }
    return *(gptr_f4)p;
}
template <int XM>
__device__ __forceinline__ void stx1(float* p, float v) {
    if (XM == XM_SCOPED) __hip_atomic_store((gptr_fw)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(gptr_fw)p = v;
}

// ---- grid barrier -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long persist_poll_scalar(const unsigned long long* p) {
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// First value of the counter that belongs to this launch.  Every launch adds exactly arrivals_per_launch, and no workgroup can
// pass the first barrier before this one arrived, so the counter read here is < base + arrivals_per_launch.
__device__ __forceinline__ unsigned long long persist_base(const PersistCtl& c, uint32_t nwg) {
    const unsigned long long per = c.arrivals_per_launch;
    const unsigned long long cur = __hip_atomic_load(c.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return cur - cur % per;
}

// Arrive: every thread of the workgroup has finished its stores (caller: s_waitcnt vmcnt(0) + __syncthreads()).  One thread.
template <int XM>
__device__ __forceinline__ void persist_arrive(const PersistCtl& c) {
    if (XM == XM_FENCE) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __hip_atomic_fetch_add(c.count, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wait (one wave, uniform): until the counter reaches target, bounded by the realtime counter so a lost co-resident workgroup
// (another kernel holding CUs) ends in an error flag, never in a hang.
template <int POLL>
__device__ __forceinline__ void persist_wait(const PersistCtl& c, unsigned long long target, bool* aborted) {
    if (*aborted) return;   // this workgroup already gave up once: keep arriving, stop waiting (the others time out on their own)
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        const unsigned long long cur = POLL == POLL_SCALAR ? persist_poll_scalar(c.count) : __hip_atomic_load(c.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((long long)(cur - target) >= 0) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > (uint64_t)c.timeout_ticks) {
            *aborted = true;   // reported by persist_report at the end of the kernel (a store here would sit in the vector-memory
            break;             // counter across the loop back edge and force a full drain before the next phase's first loads)
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ void persist_report(const PersistCtl& c, bool aborted) {
    if (aborted && (threadIdx.x & 63) == 0) __hip_atomic_store(c.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- parked rows ------------------------------------------------------------------------------------------------------------------
// The first NP rows of the workgroup's block in the NEXT phase are requested before the barrier wait into NP x KI float4 registers
// per thread (512 threads x 32 float4 = 256 KB per CU; two waves per SIMD leave 256 VGPRs each).  A first version parked them in LDS
// by LDS-DMA (global_load_lds_dwordx4): no registers held across the barrier, but the compiler's memory-counter pass treats every
// later DS access as possibly aliasing the DMA target and guarded the ring loops of the following phase with s_waitcnt vmcnt(0)
// (the stream's refills then wait a full memory latency per row); ordinary loads keep its counts exact.
// Rows [r0, r1) of a GEMV phase that workgroup wg of nwg owns (pairs stay together: RoPE / SiLU partners).
__device__ __forceinline__ void persist_rows(uint32_t M, uint32_t wg, uint32_t nwg, uint32_t* r0, uint32_t* r1) {
    const uint32_t npairs = M >> 1;
    *r0 = 2u * (uint32_t)(((uint64_t)wg * npairs) / nwg);
    *r1 = (wg + 1 == nwg) ? M : 2u * (uint32_t)(((uint64_t)(wg + 1) * npairs) / nwg);
}

// Parks rows [I0, I1) of the block.  Two calls per boundary: the EARLY rows go out as soon as the ring loop of the current phase has
// issued its last refill (they cover the epilogue and the store acknowledgement), the LATE rows after the arrive (they cover the
// barrier wait and the fetch of x).  Everything issued before the stores delays the arrive by its own landing time (the store's
// completion is observed through the same in-order counter), so the early part is kept short.
template <int KI, int NP, int I0, int I1, int MAP>
__device__ __forceinline__ void persist_park(const GemvArgs& a, const PersistCtl& c, uint32_t wg, uint32_t nwg, f4 (&pk)[NP > 0 ? NP : 1][KI], bool real = true) {
    static_assert(I0 >= 0 && I0 <= I1 && I1 <= NP, "row range");
    if (I0 == I1) return;
    const int tid = threadIdx.x;
    const uint32_t K4 = a.K >> 2;
    uint32_t r0, r1;
    persist_rows(a.M, wg, nwg, &r0, &r1);
    const char *w0 = (const char*)a.w[0], *w1 = (const char*)a.w[1], *w2 = (const char*)a.w[2];
    const uint64_t row_bytes = (uint64_t)a.K * 4;
    uint32_t loff[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) loff[j] = (uint32_t)(tid + j * PTH) < K4 ? (uint32_t)(tid + j * PTH) * 16u : 0u;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        // real == false: every register is still (re)defined, from the cache-resident dummy row - a workgroup that parks later
        // than the others must not leave the compiler a path on which last iteration's values stay live through the whole loop
        const char* p = (real && r0 + i < r1) ? gemv_row_base<MAP>(w0, w1, w2, a.rows_per_mat, r0 + i, row_bytes) : (const char*)c.dummy;
#pragma unroll
        for (int j = 0; j < KI; ++j) pk[i][j] = ld_nt_g(p + loff[j]);
    }
}
struct PersistNoEarly { __device__ __forceinline__ void operator()() const {} };

// s_waitcnt vmcnt(N) the compiler's own counter bookkeeping understands (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14])
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N < 64, "vmcnt range");
    __builtin_amdgcn_s_waitcnt((N & 0xF) | (0x7 << 4) | (0xF << 8) | ((N >> 4) << 14));
}

// ---- one GEMV phase ------------------------------------------------------------------------------------------------------------
// Same arithmetic and summation order as k_gemv_sa<KI, U, PTH, ...>: thread t owns columns 4t..4t+3 (+ 4*PTH*j), one fmaf chain per
// row walking j then x,y,z,w, DPP wave sum, cross-wave partials added in wave order.  The first NP rows of the block arrive in
// the parked registers (requested by persist_park before the barrier), the rest streams through the U-slot register ring.
template <int XM, int EPI>
__device__ __forceinline__ void persist_finish(const GemvArgs& a, const float* red, uint32_t r0, uint32_t r1, uint32_t fin, float resid_pre, double2 cs_pre,
                                               uint32_t past_pre) {
    if (r0 + fin >= r1) return;
    const float* p0 = red + fin * PNW;
    float s0 = 0.f;
#pragma unroll
    for (int k = 0; k < PNW; ++k) s0 += p0[k];
    const uint32_t v = r0 + fin;
    if (EPI == EPI_STORE) {
        stx1<XM>(a.y + v, s0);
    } else if (EPI == EPI_RESID) {
        stx1<XM>(a.y + v, __fadd_rn(s0, resid_pre));  // Add(cur, inp) ml.go:2515-2584
    } else {
        const float* p1 = p0 + PNW;
        float s1 = 0.f;
#pragma unroll
        for (int k = 0; k < PNW; ++k) s1 += p1[k];
        if (EPI == EPI_SILU_MUL) {
            stx1<XM>(a.y + (v >> 1), __fmul_rn(silu_ref(s0), s1));  // ml.go:2587-2589, 1877-1914 (llama.go:354-361)
        } else {  // EPI_QKV_ROPE (ml.go:2253-2328; cache append llama.go:274-278)
            const uint32_t d = a.d;
            if (v < 2 * d) {
                const uint32_t e = v < d ? v : v - d;
                float o0, o1;
                rope_rotate(s0, s1, cs_pre, &o0, &o1);
                // q goes to an exchange vector; the K / V rows go into the caller's cache (ordinary memory, read by other XCDs in the
                // attention phase of the same launch): agent-scope stores, whatever XM is
                if (v < d) {
                    stx1<XM>(a.q_out + e, o0);
                    stx1<XM>(a.q_out + e + 1, o1);
                } else {
                    float* dst = a.k_cache + (size_t)past_pre * d + e;
                    stx1<XM_SCOPED>(dst, o0);
                    stx1<XM_SCOPED>(dst + 1, o1);
                }
            } else {
                float* dst = a.v_cache + (size_t)past_pre * d + (v - 2 * d);
                stx1<XM_SCOPED>(dst, s0);
                stx1<XM_SCOPED>(dst + 1, s1);
            }
        }
    }
}

template <int XM, int KI, int U, int NP, int PRO, int EPI, int MAP, typename Early = PersistNoEarly>
__device__ __forceinline__ void persist_gemv(const GemvArgs& a, const PersistCtl& c, uint32_t wg, uint32_t nwg, char* smem, const f4 (&pk)[NP > 0 ? NP : 1][KI],
                                             Early early = Early()) {
    double* sred = (double*)smem;
    float* red = (float*)(smem + P_RED_OFF);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t K4 = a.K >> 2;
    uint32_t r0, r1;
    persist_rows(a.M, wg, nwg, &r0, &r1);
    const char *w0 = (const char*)a.w[0], *w1 = (const char*)a.w[1], *w2 = (const char*)a.w[2];
    const char* xdummy = (const char*)c.dummy;
    const uint32_t rpm = a.rows_per_mat;
    const uint64_t row_bytes = (uint64_t)a.K * 4;

    // Every load below is UNCONDITIONAL (inactive lanes / rows read a clamped address and discard): a load inside an exec-masked
    // branch makes the compiler fall back to s_waitcnt vmcnt(0) right there, i.e. wait for the parked rows to land before the
    // rest of x is even requested.
    f4 xr[KI];
    f4 gr[KI];
    bool act[KI];
    uint32_t loff[KI];
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        act[j] = (uint32_t)(tid + j * PTH) < K4;
        loff[j] = act[j] ? (uint32_t)(tid + j * PTH) * 16u : 0u;
        xr[j] = ldx4<XM>((const float*)((const char*)a.x + loff[j]));
        if (PRO == PRO_RMSNORM) gr[j] = *(gptr_f4)((const char*)a.gamma + loff[j]);
    }
    const uint32_t fin = (EPI == EPI_STORE || EPI == EPI_RESID) ? (uint32_t)tid : 2u * (uint32_t)tid;
    const bool fin_ok = r0 + fin < r1;
    const uint32_t vfin = fin_ok ? r0 + fin : 0u;   // row (pair) this thread finishes; row 0 always exists
    float resid_pre = 0.f;
    double2 cs_pre = double2{1.0, 0.0};
    uint32_t past_pre = 0;
    if (EPI == EPI_RESID) {
        resid_pre = ldx1<XM>(a.resid + vfin);
    } else if (EPI == EPI_QKV_ROPE) {
        past_pre = a.sp->past;
        const uint32_t e = vfin < a.d ? vfin : (vfin < 2 * a.d ? vfin - a.d : 0u);
        const __attribute__((address_space(1))) double* cs = (const __attribute__((address_space(1))) double*)(a.rope + (size_t)past_pre * (a.hd >> 1) + ((e % a.hd) >> 1));
        cs_pre = double2{cs[0], cs[1]};
    }
    // ring: the rows behind the parked ones
    const uint32_t rs = r0 + NP;
    f4 w[U][KI];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const char* p = (rs + u < r1) ? gemv_row_base<MAP>(w0, w1, w2, rpm, rs + u, row_bytes) : xdummy;
#pragma unroll
        for (int j = 0; j < KI; ++j) w[u][j] = ld_nt_g(p + loff[j]);
    }
#pragma unroll
    for (int j = 0; j < KI; ++j) {
        if (!act[j]) {
            xr[j] = f4{0.f, 0.f, 0.f, 0.f};
            if (PRO == PRO_RMSNORM) gr[j] = f4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (PRO == PRO_RMSNORM) rmsnorm_prologue<KI, PTH>(xr, act, gr, a.K, sred);
    // parked rows (requested before the barrier; they are older than x in the in-order return queue, so they are here)
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < KI; ++j) {
            const f4 cw = pk[i][j];
            s = fmaf(cw.x, xr[j].x, s);
            s = fmaf(cw.y, xr[j].y, s);
            s = fmaf(cw.z, xr[j].z, s);
            s = fmaf(cw.w, xr[j].w, s);
        }
        s = wave_sum(s);
        if (lane == 0 && r0 + i < r1) red[i * PNW + wave] = s;
    }
    for (uint32_t r = rs; r < r1; r += U) {
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t nr = r + U + u;
            const char* p = nr < r1 ? gemv_row_base<MAP>(w0, w1, w2, rpm, nr, row_bytes) : xdummy;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const f4 cw = w[u][j];
                s = fmaf(cw.x, xr[j].x, s);
                s = fmaf(cw.y, xr[j].y, s);
                s = fmaf(cw.z, xr[j].z, s);
                s = fmaf(cw.w, xr[j].w, s);
                w[u][j] = ld_nt_g(p + loff[j]);
            }
            acc[u] = s;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = wave_sum(acc[u]);
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (r + u < r1) red[(r - r0 + u) * PNW + wave] = acc[u];
        }
    }
    early();   // the next phase's early parked rows: the stream does not stop while this phase is finished
    __syncthreads();
    persist_finish<XM, EPI>(a, red, r0, r1, fin, resid_pre, cs_pre, past_pre);
    // Nothing of this phase stays in flight (the ring's tail refills, the stores): the stores must be complete before the arrive,
    // and the compiler's counter bookkeeping starts the next phase clean on every path.
    wait_vmcnt<0>();
}

// Boundary pieces for schedules where not every workgroup arrives or parks at the same point (decode: only the attention
// workgroups arrive at the barrier behind the attention phase).  `target` is the counter value that completes the barrier.
template <int XM>
__device__ __forceinline__ void persist_arrive_wg(const PersistCtl& c) {
    wait_vmcnt<0>();
    __syncthreads();
    if (threadIdx.x == 0) persist_arrive<XM>(c);
}
template <int XM, int POLL>
__device__ __forceinline__ void persist_wait_wg(const PersistCtl& c, unsigned long long target, bool* aborted) {
    if (!c.nowait && threadIdx.x < 64) persist_wait<POLL>(c, target, aborted);
    __builtin_amdgcn_s_barrier();  // raw: parked rows stay in flight across it
    if (XM == XM_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("" ::: "memory");
}

// Phase boundary: stores complete -> arrive -> park the next phase's first rows -> wait -> (acquire).
template <int XM, int POLL, int KIN, int NPN, int EN, int MAPN>
__device__ __forceinline__ void persist_boundary(const PersistCtl& c, const GemvArgs& next, uint32_t wg, uint32_t nwg, unsigned long long base, uint32_t bar_index,
                                                 bool* aborted, f4 (&pk)[NPN > 0 ? NPN : 1][KIN]) {
    wait_vmcnt<0>();
    __syncthreads();
    if (threadIdx.x == 0) persist_arrive<XM>(c);
    persist_park<KIN, NPN, EN, NPN, MAPN>(next, c, wg, nwg, pk);
    if (!c.nowait && threadIdx.x < 64) persist_wait<POLL>(c, base + (unsigned long long)(bar_index + 1) * nwg, aborted);
    __builtin_amdgcn_s_barrier();  // raw: the parked rows stay in flight across it
    if (XM == XM_FENCE) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("" ::: "memory");
}

}  // namespace lh
