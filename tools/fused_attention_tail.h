// tools/fused_attention_tail.h - NOT part of the product: the decode attention folded into the wq|wk|wv launch, as it was measured in round 4
// (profiles/r04_fused_attention_ab.txt: bit-identical to the separate launch, 6 % slower; with device-scope fences 22 % slower).  Kept so the
// experiment can be repeated: it needs, in csrc/kernels_llama.h, `uint32_t* attn_cnt; float* attn_out; float attn_scale;` at the end of GemvArgs,
// st_dev() instead of the plain q / k / v stores of gemv_finish<EPI_QKV_ROPE>, and
//     if (EPI == EPI_QKV_ROPE && a.attn_cnt) qkv_attn_tail<TH>(a, r0, r1, past_pre, smem_raw);
// after gemv_finish in k_gemv_sa and k_gemv_q8s; in csrc/plan.hip a zeroed counter array of H * QKV_TAIL_CNT_STRIDE words per plan, the three
// fields set on the "gemv_qkv_rope" launch of enqueue_decode and the k_attention launch skipped (hd == 128, ctx <= 256, 3 * embd / #CU + 2 rows
// within QKV_TAIL_SEGS segments).  tools/check_fused_attn.py is the A/B driver.
#pragma once
#include "kernels_llama.h"

namespace lh {

// Device-coherent accesses (global_store / global_load with sc1): a store is written through this XCD's L2, a load does not hit a line another
// XCD's store has made stale.  The q / k / v rows of a decode step are written and, when the attention is folded into the launch
// (qkv_attn_tail), read by workgroups of DIFFERENT XCDs inside one kernel; a release / acquire fence at device scope is a whole-L2 write-back /
// invalidate per workgroup and cost 40 us per layer when tried (profiles/r04_fused_attention_ab.txt).
__device__ __forceinline__ void st_dev(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_dev(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ f4 ld_dev4(const float* p) {
    const uint64_t lo = __hip_atomic_load((const uint64_t*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t hi = __hip_atomic_load((const uint64_t*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    f4 r;
    r.x = __builtin_bit_cast(float, (uint32_t)lo); r.y = __builtin_bit_cast(float, (uint32_t)(lo >> 32));
    r.z = __builtin_bit_cast(float, (uint32_t)hi); r.w = __builtin_bit_cast(float, (uint32_t)(hi >> 32));
    return r;
}


// ---------------------------------------------------------------------------------------------------
// Decode attention WITHOUT a launch (round 4).  The wq|wk|wv launch owns the 3*hd rows of head h (its q, the new k and the new v)
// in a handful of workgroups; each workgroup waits for its write-through stores, adds its row count to the head's counter, and the
// workgroup whose add completes the count computes the head (device-coherent loads, exactly k_attention's arithmetic).  Nobody waits for
// anybody: a workgroup that is not last simply ends.  attn_head_tail reproduces k_attention (1024 threads: 32 key groups, 8 key phases
// in the PV step, 16 waves in the long-row reductions) with TH_ threads by giving every thread 1024 / TH_ of those roles and keeping each
// role's summation order, so the result is bit-identical to the separate launch (and to the batched ticks, which keep it).
// hd = 128 only (the plan decides).
// ---------------------------------------------------------------------------------------------------
template <int TH_>
__device__ __forceinline__ void attn_head_tail(const float* q_all, const float* kc, const float* vc, float* out, uint32_t d, uint32_t h, uint32_t T,
                                               float scale, char* smem) {
    constexpr int HD = 128, NG = TH_ / 32, NWV = TH_ / 64, PPT = ATT_TH / TH_, PHASES = ATT_TH / HD, VWAVES = ATT_TH / 64;
    static_assert(TH_ % HD == 0 && ATT_TH % TH_ == 0, "attn_head_tail: workgroup size");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t Tp = (T + 63) & ~63u;
    float* sc = (float*)smem;         // [Tp] scaled scores
    float* pr = sc + Tp;              // [Tp] un-normalised probabilities
    float* scratch = pr + Tp;         // [ATT_TH] PV partials / reduction scratch
    const float* q = q_all + h * HD;
    const float* Kc = kc + h * HD;
    const float* Vc = vc + h * HD;
    const uint32_t c = tid % HD, tr = tid / HD;   // this thread plays key phases tr * PPT .. + PPT - 1 of column c
    constexpr int VP = 8;
    float vpre[PPT][VP];
#pragma unroll
    for (int s = 0; s < PPT; ++s)
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const uint32_t t = tr * PPT + s + (uint32_t)i * PHASES;
            vpre[s][i] = t < T ? ld_dev(Vc + (size_t)t * d + c) : 0.f;
        }
    {   // scores: one key per 32-lane group, 128 keys requested per round
        const int g = tid >> 5, gl = tid & 31;
        constexpr int UN = 128 / NG;
        const f4 qv = ld_dev4(q + gl * 4);
        for (uint32_t t0 = g; t0 < T; t0 += NG * UN) {
            f4 kv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t t = t0 + u * NG;
                kv[u] = ld_dev4(Kc + (size_t)(t < T ? t : 0) * d + gl * 4);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t t = t0 + u * NG;
                float s = fmaf(kv[u].x, qv.x, 0.f);
                s = fmaf(kv[u].y, qv.y, s); s = fmaf(kv[u].z, qv.z, s); s = fmaf(kv[u].w, qv.w, s);
                s = half_wave_sum(s);
                if (gl == 0 && t < T) sc[t] = __fmul_rn(s, scale);
            }
        }
    }
    __syncthreads();
    float inv;
    if (T <= 128) {   // every wave evaluates the row redundantly (k_attention's short-row path)
        float m = -INFINITY;
        for (uint32_t t = lane; t < T; t += 64) m = fmaxf(m, sc[t]);
        m = wave_max(m);
        float psum = 0.f;
        for (uint32_t t = lane; t < T; t += 64) {
            const float p = (float)exp((double)__fsub_rn(sc[t], m));
            pr[t] = p;
            psum += p;
        }
        psum = wave_sum(psum);
        inv = __fdiv_rn(1.0f, psum);
    } else {          // k_attention's long-row path: thread v of 1024 takes keys v, v + 1024, ..; wave sums added in wave order
        float mv[PPT];
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            float m = -INFINITY;
            for (uint32_t t = tid + r * TH_; t < T; t += ATT_TH) m = fmaxf(m, sc[t]);
            mv[r] = wave_max(m);
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < PPT; ++r) scratch[wave + r * NWV] = mv[r];
        }
        __syncthreads();
        float m = scratch[0];
#pragma unroll
        for (int w = 1; w < VWAVES; ++w) m = fmaxf(m, scratch[w]);
        __syncthreads();
        float ps[PPT];
#pragma unroll
        for (int r = 0; r < PPT; ++r) {
            float psum = 0.f;
            for (uint32_t t = tid + r * TH_; t < T; t += ATT_TH) {
                const float p = (float)exp((double)__fsub_rn(sc[t], m));
                pr[t] = p;
                psum += p;
            }
            ps[r] = wave_sum(psum);
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < PPT; ++r) scratch[wave + r * NWV] = ps[r];
        }
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < VWAVES; ++w) tot += scratch[w];
        inv = __fdiv_rn(1.0f, tot);
        __syncthreads();
    }
    float acc[PPT];
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        acc[s] = 0.f;
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const uint32_t t = tr * PPT + s + (uint32_t)i * PHASES;
            if (t < T) acc[s] = fmaf(vpre[s][i], __fmul_rn(pr[t], inv), acc[s]);
        }
    }
    for (uint32_t base = VP * PHASES; base < T; base += VP * PHASES) {   // keys beyond the first 64: the same batches of VP per phase
        float vv[PPT][VP];
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int i = 0; i < VP; ++i) {
                const uint32_t t = base + tr * PPT + s + (uint32_t)i * PHASES;
                vv[s][i] = ld_dev(Vc + (size_t)(t < T ? t : 0) * d + c);
            }
#pragma unroll
        for (int s = 0; s < PPT; ++s)
#pragma unroll
            for (int i = 0; i < VP; ++i) {
                const uint32_t t = base + tr * PPT + s + (uint32_t)i * PHASES;
                if (t < T) acc[s] = fmaf(vv[s][i], __fmul_rn(pr[t], inv), acc[s]);
            }
    }
#pragma unroll
    for (int s = 0; s < PPT; ++s) scratch[(tr * PPT + s) * HD + c] = acc[s];
    __syncthreads();
    if (tid < HD) {
        float o = scratch[tid];
#pragma unroll
        for (int p2 = 1; p2 < PHASES; ++p2) o += scratch[tid + p2 * HD];
        out[h * HD + tid] = o;
    }
}

constexpr int QKV_TAIL_CNT_STRIDE = 64;   // one arrival counter per 256 bytes: the adds of different heads do not queue on one line
constexpr int QKV_TAIL_SEGS = 8;   // (matrix, head) runs a workgroup's block of virtual rows can touch

template <int TH_>
__device__ __forceinline__ void qkv_attn_tail(const GemvArgs& a, uint32_t r0, uint32_t r1, uint32_t past, char* smem) {
    const int tid = threadIdx.x;
    const uint32_t d = a.d, hd = a.hd;
    // the q / k / v values were stored write-through (st_dev): once the stores have completed they are visible device-wide.  Wait for them
    // (a workgroup-scope release does not: within a CU program order is enough), then count; no L2 write-back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t* done = (uint32_t*)(smem + 16384);   // [QKV_TAIL_SEGS]: head completed by this workgroup's add, or ~0 (behind attn_head_tail's 6 KB)
    if (tid < QKV_TAIL_SEGS) {
        // segment `tid` of [r0, r1): cut at every multiple of hd (matrix boundaries are multiples of hd)
        const uint32_t v0 = tid == 0 ? r0 : (r0 / hd + (uint32_t)tid) * hd;
        uint32_t v1 = (r0 / hd + (uint32_t)tid + 1) * hd;
        if (v1 > r1) v1 = r1;
        uint32_t res = ~0u;
        if (v0 < r1) {
            const uint32_t h = (v0 % d) / hd, cnt = v1 - v0;
            const uint32_t old = __hip_atomic_fetch_add(a.attn_cnt + h * QKV_TAIL_CNT_STRIDE, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + cnt == 3 * hd) {
                __hip_atomic_store(a.attn_cnt + h * QKV_TAIL_CNT_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next touched by the next launch
                res = h;
            }
        }
        done[tid] = res;
    }
    __syncthreads();
#pragma nounroll
    for (int i = 0; i < QKV_TAIL_SEGS; ++i) {
        const uint32_t h = done[i];
        if (h == ~0u) continue;            // uniform over the workgroup
        attn_head_tail<TH_>(a.q_out, a.k_cache, a.v_cache, a.attn_out, d, h, past + 1, a.attn_scale, smem);
        __syncthreads();
    }
}

}  // namespace lh
