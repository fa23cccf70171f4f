// tools/attn_latency_probe.hip — what the decode attention launch (k_attention: one 1024-thread workgroup per head) costs at short contexts, against the floor of
// launching that shape at all: empty kernels of 32 x 1024 / 32 x 256 threads, a kernel that only follows the dependent loads (args -> position -> one cache row),
// and k_attention itself at T = 16 / 64 / 128.  Back-to-back launches on one stream, timing only.  Not product code.
#include "../llama.go_amd/csrc/kernels_llama.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
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
    printf("done\n");
    return 0;
}
