// tools/attn_latency_probe.hip — what the decode attention launch (k_attention: one 1024-thread workgroup per head) costs at short contexts, against the floor of
// launching that shape at all: empty kernels of 32 x 1024 / 32 x 256 threads, a kernel that only follows the dependent loads (args -> position -> one cache row),
// and k_attention itself at T = 16 / 64 / 128.  Back-to-back launches on one stream, timing only.  Not product code.
#include "../llama.go_amd/csrc/kernels_llama.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
namespace lh {
__global__ __launch_bounds__(ATT_TH) void k_attention_traced(const AttnArgs a, unsigned long long* tr) {
    unsigned long long st_[10]; int ns_ = 0;
#define ATT_STAMP() do { st_[ns_++] = __builtin_amdgcn_s_memtime(); } while (0)
    ATT_STAMP();
    LH_TOUCH_ARGS(a.q, a.sp, a.rows);   // both lines of the argument block at once (rows -> sp -> position was three dependent scalar misses)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NWV = ATT_TH / 64, NG = ATT_TH / 32;  // waves, 32-lane key groups
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t h = blockIdx.x, j = blockIdx.y;
    const uint32_t past = a.rows ? a.rows[j].pos : (a.sp ? a.sp->past : a.past_host);
    const uint32_t T = a.rows ? past + 1 : past + j + 1;  // keys 0..past+j are visible to query j (mask: i > past + j, ml.go:2401-2404)
    ATT_STAMP();   // 1: position known
    const uint32_t Tp = (T + 63) & ~63u;
    float* sc = (float*)smem_raw;     // [Tp] scaled scores
    float* pr = sc + Tp;              // [Tp] un-normalised probabilities
    float* scratch = pr + Tp;         // [ATT_TH] PV partials / reduction scratch
    const uint32_t d = a.d, hd = a.hd;
    const float* q = a.q + (size_t)j * d + h * hd;
    // (the cache pointers come out of a select between a kernel argument and a pointer read from the row table: say that they are GLOBAL memory, or every
    // K / V load is a flat_load - counted on the LDS counter too, so each wait for cache rows also drained the LDS queue; round 6, ISA of this kernel)
    typedef const float __attribute__((address_space(1))) gfl;
    typedef const f4 __attribute__((address_space(1))) gf4;
    gfl* Kc = (gfl*)(uintptr_t)((a.rows ? a.rows[j].kc + a.kv_off : a.k_cache) + h * hd);
    gfl* Vc = (gfl*)(uintptr_t)((a.rows ? a.rows[j].vc + a.kv_off : a.v_cache) + h * hd);
    // The cache rows of one head are 512 B segments strided by embd: every loop below keeps several INDEPENDENT row
    // loads in flight per lane (a dependent one-row-per-iteration loop costs a full L2 latency per key: 0.27 us/key measured).
    const uint32_t phases = ATT_TH / hd;  // hd = 128 -> 8 key phases in the PV step
    const uint32_t c = tid % hd, ph = tid / hd;
    constexpr int VP = 8;
    float vpre[VP];
#pragma unroll
    for (int i = 0; i < VP; ++i) {   // first V rows: issued before anything else, consumed last
        const uint32_t t = ph + (uint32_t)i * phases;
        vpre[i] = t < T ? Vc[(size_t)t * d + c] : 0.f;
    }
    // --- scores: one key per 32-lane group, UN keys in flight per group (hd = 128 -> float4 per lane; other hd: strided loop)
    const int g = tid >> 5, gl = tid & 31;
    constexpr int UN = 4;
    if (hd == 128) {
        const f4 qv = *(const f4*)(q + gl * 4);
        for (uint32_t t0 = g; t0 < T; t0 += NG * UN) {
            f4 kv[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t t = t0 + u * NG;
                kv[u] = *(gf4*)(Kc + (size_t)(t < T ? t : 0) * d + gl * 4);
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
                const f4 kv = *(gf4*)(Kc + (size_t)t * d + cc);
                const f4 qv = *(const f4*)(q + cc);
                s = fmaf(kv.x, qv.x, s); s = fmaf(kv.y, qv.y, s); s = fmaf(kv.z, qv.z, s); s = fmaf(kv.w, qv.w, s);
            }
            s = half_wave_sum(s);
            if (gl == 0) sc[t] = __fmul_rn(s, a.scale);
        }
    }
    ATT_STAMP();   // 2: scores written
    __syncthreads();
    ATT_STAMP();   // 3: behind barrier 1
    // --- softmax (ml.go:2432-2505): max, p = fl32(exp_f64(fl32(s - max))), fp32 sum, p *= 1/sum
    float inv;
    if (T <= 128) {
        // short rows: every wave evaluates the whole row redundantly with wave-level reductions (same code -> same bits):
        // no block barrier in this phase; waves only read sc[] and write identical values to pr[]
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
        // long rows: the f64 exps are spread over all threads, two block reductions in fixed order
        float m = -INFINITY;
        for (uint32_t t = tid; t < T; t += ATT_TH) m = fmaxf(m, sc[t]);
        m = wave_max(m);
        if (lane == 0) scratch[wave] = m;
        __syncthreads();
        m = scratch[0];
#pragma unroll
        for (int w = 1; w < NWV; ++w) m = fmaxf(m, scratch[w]);
        __syncthreads();
        float psum = 0.f;
        for (uint32_t t = tid; t < T; t += ATT_TH) {
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
    ATT_STAMP();   // 4: softmax done
    // --- PV: thread (c, ph) accumulates its key phase, VP independent row loads in flight.  (T <= 128: DS operations of a
    // wave execute in order, so pr[] written above by this wave is visible to its own reads; T > 128: barrier above.)
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
            vv[i] = Vc[(size_t)(t < T ? t : 0) * d + c];
        }
#pragma unroll
        for (int i = 0; i < VP; ++i) {
            const uint32_t t = t0 + (uint32_t)i * phases;
            if (t < T) acc = fmaf(vv[i], __fmul_rn(pr[t], inv), acc);
        }
    }
    ATT_STAMP();   // 5: PV done
    scratch[tid] = acc;
    __syncthreads();
    ATT_STAMP();   // 6: behind barrier 2
    if (tid < (int)hd) {
        float o = scratch[tid];
        for (uint32_t p2 = 1; p2 < phases; ++p2) o += scratch[tid + p2 * hd];
        a.out[(size_t)j * d + h * hd + tid] = o;
        if (a.out_s3) attn_store_split3(a, (size_t)j * d + h * hd + tid, o);
    }
    ATT_STAMP();   // 7: stored
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) for (int i = 0; i < ns_; ++i) tr[i] = st_[i];
}
}  // namespace lh
__global__ void k_empty(float* out) { if (threadIdx.x == 9999) out[0] = 1.f; }
__global__ void k_chain(const StepParams* sp, const float* kc, float* out, uint32_t d) {
    const uint32_t past = sp->past;
    const float v = kc[(size_t)past * d + blockIdx.x * 128 + (threadIdx.x & 127)];
    if (threadIdx.x < 128) out[blockIdx.x * 128 + threadIdx.x] = v;
}
int main() {
    CK(hipSetDevice(0));
    const uint32_t d = 4096, H = 32, ctx = 128;
    float *q, *kc, *vc, *out; StepParams* sp;
    CK(hipMalloc(&q, d * 4)); CK(hipMalloc(&kc, (size_t)ctx * d * 4)); CK(hipMalloc(&vc, (size_t)ctx * d * 4)); CK(hipMalloc(&out, d * 4)); CK(hipMalloc(&sp, sizeof(StepParams)));
    std::vector<float> h((size_t)ctx * d); unsigned s = 7; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    CK(hipMemcpy(kc, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(vc, h.data(), h.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(q, h.data(), d * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // a chain of 400 launches captured into ONE hipGraph (as the decode step is): eager launches are bound by the host at ~2.4 us each
    auto timeit = [&](const char* label, auto launch) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 400; ++i) launch();
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); const double us = ms * 1e3 / 400; best = us < best ? us : best;
        }
        printf("  %-70s %7.2f us\n", label, best); CK(hipGetLastError());
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    };
    timeit("empty kernel, 32 workgroups x 1024 threads", [&] { hipLaunchKernelGGL(k_empty, dim3(32), dim3(1024), 0, st, out); });
    timeit("empty kernel, 32 workgroups x 256 threads", [&] { hipLaunchKernelGGL(k_empty, dim3(32), dim3(256), 0, st, out); });
    timeit("empty kernel, 256 workgroups x 256 threads", [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, st, out); });
    timeit("dependent loads only (position -> one cache row -> store), 32 x 256", [&] { hipLaunchKernelGGL(k_chain, dim3(32), dim3(256), 0, st, sp, kc, out, d); });
    for (uint32_t past : {15u, 63u, 127u}) {
        StepParams hsp = {1, past, 0, 0}; CK(hipMemcpy(sp, &hsp, sizeof hsp, hipMemcpyHostToDevice));
        AttnArgs a = {};
        a.q = q; a.k_cache = kc; a.v_cache = vc; a.out = out; a.d = d; a.hd = 128; a.n = 1; a.scale = 0.088388f; a.sp = sp;
        const size_t lds = (2 * (size_t)((ctx + 63) & ~63u) + ATT_TH) * 4;
        char label[128]; snprintf(label, sizeof label, "k_attention, T = %u (32 x 1024 threads)", past + 1);
        timeit(label, [&] { hipLaunchKernelGGL(k_attention, dim3(H, 1), dim3(ATT_TH), lds, st, a); });
    }
    {   // phase stamps of one workgroup (s_memtime, 100 MHz-independent shader-clock counter): where the ~2 us of the kernel body go at T = 16
        unsigned long long* tr; CK(hipMalloc(&tr, 16 * 8));
        StepParams hsp = {1, 15, 0, 0}; CK(hipMemcpy(sp, &hsp, sizeof hsp, hipMemcpyHostToDevice));
        AttnArgs a = {};
        a.q = q; a.k_cache = kc; a.v_cache = vc; a.out = out; a.d = d; a.hd = 128; a.n = 1; a.scale = 0.088388f; a.sp = sp;
        const size_t lds = (2 * (size_t)((ctx + 63) & ~63u) + ATT_TH) * 4;
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_attention_traced, dim3(H, 1), dim3(ATT_TH), lds, st, a, tr);
        CK(hipStreamSynchronize(st));
        unsigned long long h[16]; CK(hipMemcpy(h, tr, sizeof h, hipMemcpyDeviceToHost));
        const char* names[] = {"position known", "scores written", "behind barrier 1", "softmax done", "PV done", "behind barrier 2", "stored"};
        printf("  k_attention at T = 16, wave 0 of head 0, s_memtime ticks since the kernel's first instruction (100 MHz constant clock: 1 tick = 10 ns):\n");
        for (int i = 1; i < 8; ++i) printf("    %-18s %6llu\n", names[i - 1], h[i] - h[0]);
    }
    printf("done\n");
    return 0;
}
