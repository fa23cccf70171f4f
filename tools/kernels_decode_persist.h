// MOVED OUT OF THE PRODUCT in round 3 (was csrc/): the resident-kernel experiment measured a tie with the per-layer kernels and produced one
// unexplained wrong token (profiles/r02b_resident_trace.txt, r02c_resident_trace_again.txt); kept only for tools/persist_probe.hip / resident_probe.hip.
// csrc/kernels_decode_persist.h — one decode step (llama.Eval with N = 1, pkg/llama/llama.go:246-384) as ONE resident kernel:
// a workgroup per CU walks  qkv_rope | attention | wo_resid | w1w3_silu | w2_resid  for every layer and the lm_head, with a grid
// barrier between the phases (csrc/kernels_persist.h).  Against the five-launches-per-layer plan it removes
//   * the attention launch: the 4.9 us kernel (32 workgroups, latency-bound) runs on the first H workgroups WHILE every workgroup's
//     block of wo (16 rows = 256 KB per CU at 7B: the whole matrix) is already streaming into registers - wo does not depend on
//     anything the attention computes;
//   * the ramp at the head of every GEMV: the first rows of the next matrix are requested before the barrier wait.
// The activation vectors handed from phase to phase live in UNCACHED device memory (plain loads and stores are coherent across the
// eight XCD L2s without cache maintenance: tools/persist_probe, profiles/r02b_persist_probe_run1.txt: fences 209-285 us per layer,
// uncached 127); the K / V rows of the step go into the caller's ordinary cache with agent-scope stores and are read back with
// agent-scope loads.  Arithmetic and summation order per GEMV are those of k_gemv_sa at 512 threads.
#pragma once
#include "kernels_persist.h"

namespace lh {

struct PersistLayer {
    const float *attn_norm, *wq, *wk, *wv, *wo, *ffn_norm, *w1, *w3, *w2;
    float *kc, *vc;                // this layer's cache slot [ctx][d]
};

struct PersistDecodeArgs {
    const PersistLayer* layers;    // [n_layers], ordinary device memory
    uint32_t n_layers;
    uint32_t d, F, V, H, hd;
    float *xa, *xb, *q, *attn, *g; // exchange vectors (uncached): residual ping-pong [d], roped q [d], merged heads [d], gated ff [F]
    const float *norm, *output;    // final RMSNorm weight, lm_head matrix [V][d]
    float* logits;                 // [V], ordinary memory (read by the next kernel)
    const double2* rope;
    const StepParams* sp;
    float scale;                   // fl32(1/sqrt(hd)) llama.go:306
    PersistCtl ctl;
};

// ---- attention of one head for the single query of a decode step (llama.go:300-333), PTH threads --------------------------------
// Same scheme as k_attention (scores per 32-lane key group, softmax as ml.go:2432-2505 with f64 exp, PV per (column, key phase)),
// loads of cache rows agent-scoped, output to the uncached exchange vector.  LDS: sc[Tp] | pr[Tp] | scratch[PTH].
__device__ __forceinline__ void persist_attention(const PersistDecodeArgs& a, const float* kcache, const float* vcache, uint32_t h, char* smem) {
    constexpr int TH = PTH, NWV = TH / 64, NG = TH / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t past = a.sp->past;
    const uint32_t T = past + 1;
    const uint32_t Tp = (T + 63) & ~63u;
    float* sc = (float*)smem;
    float* pr = sc + Tp;
    float* scratch = pr + Tp;
    const uint32_t d = a.d, hd = a.hd;
    const float* q = a.q + h * hd;
    const float* Kc = kcache + h * hd;
    const float* Vc = vcache + h * hd;
    const uint32_t phases = TH / hd;
    const uint32_t c = tid % hd, ph = tid / hd;
    constexpr int VP = 8;
    float vpre[VP];
#pragma unroll
    for (int i = 0; i < VP; ++i) {   // first V rows: issued before anything else, consumed last
        const uint32_t t = ph + (uint32_t)i * phases;
        vpre[i] = ldx1<XM_SCOPED>(Vc + (size_t)(t < T ? t : 0) * d + c);
    }
    const int g = tid >> 5, gl = tid & 31;
    constexpr int UN = 4;
    if (hd == 128) {
        const f4 qv = *(gptr_f4)(q + gl * 4);
        for (uint32_t t0 = g; t0 < T; t0 += NG * UN) {
            f4 kv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t t = t0 + u * NG;
                kv[u] = ldx4<XM_SCOPED>(Kc + (size_t)(t < T ? t : 0) * d + gl * 4);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t t = t0 + u * NG;
                float s = fmaf(kv[u].x, qv.x, 0.f);
                s = fmaf(kv[u].y, qv.y, s); s = fmaf(kv[u].z, qv.z, s); s = fmaf(kv[u].w, qv.w, s);
                s = half_wave_sum(s);
                if (gl == 0 && t < T) sc[t] = __fmul_rn(s, a.scale);  // Scale ml.go:2331-2374
            }
        }
    } else {
        for (uint32_t t = g; t < T; t += NG) {
            float s = 0.f;
            for (uint32_t cc = gl * 4; cc < hd; cc += 128) {
                const f4 kv = ldx4<XM_SCOPED>(Kc + (size_t)t * d + cc);
                const f4 qv = *(gptr_f4)(q + cc);
                s = fmaf(kv.x, qv.x, s); s = fmaf(kv.y, qv.y, s); s = fmaf(kv.z, qv.z, s); s = fmaf(kv.w, qv.w, s);
            }
            s = half_wave_sum(s);
            if (gl == 0) sc[t] = __fmul_rn(s, a.scale);
        }
    }
    __syncthreads();
    // softmax (ml.go:2432-2505): max, p = fl32(exp_f64(fl32(s - max))), fp32 sum, p *= 1/sum
    float inv;
    if (T <= 128) {
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
    } else {
        float m = -INFINITY;
        for (uint32_t t = tid; t < T; t += TH) m = fmaxf(m, sc[t]);
        m = wave_max(m);
        if (lane == 0) scratch[wave] = m;
        __syncthreads();
        m = scratch[0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) m = fmaxf(m, scratch[w]);
        __syncthreads();
        float psum = 0.f;
        for (uint32_t t = tid; t < T; t += TH) {
            const float p = (float)exp((double)__fsub_rn(sc[t], m));
            pr[t] = p;
            psum += p;
        }
        psum = wave_sum(psum);
        if (lane == 0) scratch[wave] = psum;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) tot += scratch[w];
        inv = __fdiv_rn(1.0f, tot);
        __syncthreads();
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < VP; ++i) {
        const uint32_t t = ph + (uint32_t)i * phases;
        if (t < T) acc = fmaf(vpre[i], __fmul_rn(pr[t], inv), acc);
    }
    for (uint32_t t0 = ph + VP * phases; t0 < T; t0 += VP * phases) {
        float vv[VP];
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const uint32_t t = t0 + (uint32_t)i * phases;
            vv[i] = ldx1<XM_SCOPED>(Vc + (size_t)(t < T ? t : 0) * d + c);
        }
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const uint32_t t = t0 + (uint32_t)i * phases;
            if (t < T) acc = fmaf(vv[i], __fmul_rn(pr[t], inv), acc);
        }
    }
    scratch[tid] = acc;
    __syncthreads();
    if (tid < (int)hd) {
        float o = scratch[tid];
        for (uint32_t p2 = 1; p2 < phases; ++p2) o += scratch[tid + p2 * hd];
        *(gptr_fw)(a.attn + h * hd + tid) = o;
    }
}

// Rows parked across a barrier per phase, by float4-per-thread count of the phase (KI): about 32 float4 (128 VGPRs) at most.
#ifndef PERSIST_NP2
#define PERSIST_NP2 8
#endif
#ifndef PERSIST_NP6
#define PERSIST_NP6 3
#endif
__host__ __device__ constexpr int persist_np(int ki) { return ki <= 1 ? 16 : ki == 2 ? PERSIST_NP2 : ki == 3 ? 4 : ki == 4 ? 3 : ki <= 6 ? PERSIST_NP6 : ki <= 8 ? 2 : 1; }
// wo streams under the attention: park deep.  (KI = 2: 16 rows = all of a 7B workgroup's block would be 128 VGPRs on top of the
// GEMV's working set and spills; scratch reloads are vector-memory operations and would queue behind the parked rows.)
__host__ __device__ constexpr int persist_np_wo(int ki) { return ki <= 1 ? 16 : ki == 2 ? 12 : ki == 3 ? 5 : 3; }
__host__ __device__ constexpr int persist_u(int ki) { return ki <= 3 ? 2 : 1; }

template <int KI_D, int KI_F>
__global__ __launch_bounds__(PTH) void k_decode_persist(const PersistDecodeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* smem = smem_raw;
    constexpr int XM = XM_PLAIN, POLL = POLL_VECTOR;
    constexpr int NPD = persist_np(KI_D), NPO = persist_np_wo(KI_D), NPF = persist_np(KI_F), UD = persist_u(KI_D), UF = persist_u(KI_F);
    const uint32_t wg = blockIdx.x, nwg = gridDim.x;
    const PersistCtl& c = a.ctl;
    unsigned long long target = persist_base(c, nwg);
    bool aborted = false;
    uint32_t stamp = 0;
    (void)stamp;
    const uint32_t d = a.d, F = a.F, H = a.H;
    auto qkv_args = [&](const PersistLayer& L) {
        GemvArgs g = {};
        g.w[0] = L.wq; g.w[1] = L.wk; g.w[2] = L.wv; g.rows_per_mat = d; g.M = 3 * d; g.K = d; g.x = a.xa; g.gamma = L.attn_norm;
        g.q_out = a.q; g.k_cache = L.kc; g.v_cache = L.vc; g.rope = a.rope; g.hd = a.hd; g.d = d; g.sp = a.sp;
        return g;
    };
    auto lm_args = [&]() {   // MAP_BLOCK with all rows in "matrix 0": the same instantiation of the parking code as the qkv phase
        GemvArgs g = {};
        g.w[0] = a.output; g.w[1] = a.output; g.w[2] = a.output; g.rows_per_mat = a.V; g.M = a.V; g.K = d; g.x = a.xa; g.gamma = a.norm; g.y = a.logits;
        return g;
    };
    f4 pkD[NPD][KI_D], pkO[NPO][KI_D], pkF[NPF][KI_F];
    {
        const GemvArgs g0 = qkv_args(a.layers[0]);
        persist_park<KI_D, NPD, 0, NPD, MAP_BLOCK>(g0, c, wg, nwg, pkD);
    }
    for (uint32_t il = 0; il < a.n_layers; ++il) {
        const PersistLayer L = a.layers[il];
        PERSIST_STAMP(c, wg, nwg, stamp);   // 0: layer start
        {   // RMSNorm*gamma -> wq|wk|wv -> RoPE(Q, new K) -> K,V appended to the cache   (llama.go:255-297)
            const GemvArgs g = qkv_args(L);
            persist_gemv<XM, KI_D, UD, NPD, PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK>(g, c, wg, nwg, smem, pkD);
        }
        PERSIST_STAMP(c, wg, nwg, stamp);   // 1: qkv done
        GemvArgs gwo = {};
        gwo.w[0] = L.wo; gwo.M = d; gwo.K = d; gwo.x = a.attn; gwo.resid = a.xa; gwo.y = a.xb;
        // barrier 1 (everybody): q and the new K / V rows are out.  Workgroups without a head start streaming their block of wo now.
        persist_arrive_wg<XM>(c);
        persist_park<KI_D, NPO, 0, NPO, MAP_SINGLE>(gwo, c, wg, nwg, pkO, wg >= H);   // head workgroups: after their attention (below)
        target += nwg;
        persist_wait_wg<XM, POLL>(c, target, &aborted);
        PERSIST_STAMP(c, wg, nwg, stamp);   // 2: barrier 1 passed
        // attention on the first H workgroups (llama.go:300-333); barrier 2 counts only their arrivals
        if (wg < H) {
            persist_attention(a, L.kc, L.vc, wg, smem);
            persist_arrive_wg<XM>(c);
            persist_park<KI_D, NPO, 0, NPO, MAP_SINGLE>(gwo, c, wg, nwg, pkO);
        }
        target += H;
        persist_wait_wg<XM, POLL>(c, target, &aborted);
        PERSIST_STAMP(c, wg, nwg, stamp);   // 3: attention + barrier 2
        // wo + residual   (llama.go:336-340)
        persist_gemv<XM, KI_D, UD, NPO, PRO_PLAIN, EPI_RESID, MAP_SINGLE>(gwo, c, wg, nwg, smem, pkO);
        PERSIST_STAMP(c, wg, nwg, stamp);   // 4: wo done
        {   // RMSNorm*gamma -> w1|w3 -> silu(w1 h) * (w3 h)   (llama.go:346-361)
            GemvArgs g = {};
            g.w[0] = L.w1; g.w[1] = L.w3; g.M = 2 * F; g.K = d; g.x = a.xb; g.gamma = L.ffn_norm; g.y = a.g;
            persist_arrive_wg<XM>(c);
            persist_park<KI_D, NPD, 0, NPD, MAP_PAIR>(g, c, wg, nwg, pkD);
            target += nwg;
            persist_wait_wg<XM, POLL>(c, target, &aborted);
            PERSIST_STAMP(c, wg, nwg, stamp);   // 5: barrier 3
            persist_gemv<XM, KI_D, UD, NPD, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>(g, c, wg, nwg, smem, pkD);
            PERSIST_STAMP(c, wg, nwg, stamp);   // 6: w1w3 done
        }
        {   // w2 + residual   (llama.go:363-366)
            GemvArgs g = {};
            g.w[0] = L.w2; g.M = d; g.K = F; g.x = a.g; g.resid = a.xb; g.y = a.xa;
            persist_arrive_wg<XM>(c);
            persist_park<KI_F, NPF, 0, NPF, MAP_SINGLE>(g, c, wg, nwg, pkF);
            target += nwg;
            persist_wait_wg<XM, POLL>(c, target, &aborted);
            PERSIST_STAMP(c, wg, nwg, stamp);   // 7: barrier 4
            persist_gemv<XM, KI_F, UF, NPF, PRO_PLAIN, EPI_RESID, MAP_SINGLE>(g, c, wg, nwg, smem, pkF);
            PERSIST_STAMP(c, wg, nwg, stamp);   // 8: w2 done
        }
        {   // next: the following layer's qkv, or the lm_head
            const bool last = il + 1 == a.n_layers;
            const GemvArgs g = last ? lm_args() : qkv_args(a.layers[last ? il : il + 1]);
            persist_arrive_wg<XM>(c);
            persist_park<KI_D, NPD, 0, NPD, MAP_BLOCK>(g, c, wg, nwg, pkD);
            target += nwg;
            persist_wait_wg<XM, POLL>(c, target, &aborted);
            PERSIST_STAMP(c, wg, nwg, stamp);   // 9: barrier 5
        }
    }
    {   // final RMSNorm*gamma -> lm_head   (llama.go:374-384)
        const GemvArgs g = lm_args();
        persist_gemv<XM, KI_D, UD, NPD, PRO_RMSNORM, EPI_STORE, MAP_BLOCK>(g, c, wg, nwg, smem, pkD);
        PERSIST_STAMP(c, wg, nwg, stamp);   // lm_head done
    }
    persist_report(c, aborted);
}

// arrivals one launch adds to the counter (PersistCtl::arrivals_per_launch)
__host__ __device__ inline unsigned long long persist_decode_arrivals(uint32_t n_layers, uint32_t nwg, uint32_t H) { return (unsigned long long)n_layers * (4ull * nwg + H); }

}  // namespace lh
