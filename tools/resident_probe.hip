// tools/resident_probe.hip — the PRODUCT resident decode kernel (csrc/kernels_decode_persist.h) on a synthetic 7B-shape model, standalone
// (no host library, no torch): launch time, and with -DPERSIST_TRACE per-phase realtime stamps of three workgroups (first = an attention
// workgroup, middle, last) averaged over the layers.  Not product code.
#ifndef PERSIST_TRACE
#define PERSIST_TRACE
#endif
#include "kernels_decode_persist.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void k_fill(float* p, size_t n, uint32_t seed, float scale, float offset) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        uint64_t z = (i + 0x9E3779B97F4A7C15ull * (seed + 1));
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = offset + ((float)(int)((z >> 40) & 0xFFFFFF) - 8388608.0f) * (1.0f / 8388608.0f) * scale;
    }
}

int main(int argc, char** argv) {
    const uint32_t L = argc > 1 ? atoi(argv[1]) : 32, past = argc > 2 ? atoi(argv[2]) : 20;
    const uint32_t d = 4096, F = 11008, V = 32000, H = 32, hd = 128, ctx = 128;
    CK(hipSetDevice(0)); hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); const int nCU = pr.multiProcessorCount;
    hipStream_t st; CK(hipStreamCreate(&st)); hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t per_layer = (size_t)4 * d * d + (size_t)3 * d * F + 2 * d;
    float* W; CK(hipMalloc(&W, (per_layer * L + (size_t)V * d + d) * 4));
    std::vector<PersistLayer> hl(L);
    float *kc, *vc; CK(hipMalloc(&kc, (size_t)L * ctx * d * 4)); CK(hipMalloc(&vc, (size_t)L * ctx * d * 4));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, kc, (size_t)L * ctx * d, 91u, 0.5f, 0.f);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, st, vc, (size_t)L * ctx * d, 92u, 0.5f, 0.f);
    size_t off = 0; uint32_t seed = 0;
    auto mat = [&](size_t rows, size_t cols) { float* p = W + off; hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, p, rows * cols, seed++, sqrtf(3.0f / cols), 0.f); off += rows * cols; return (const float*)p; };
    auto vec = [&](size_t n) { float* p = W + off; hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, st, p, n, seed++, 0.1f, 1.0f); off += n; return (const float*)p; };
    for (uint32_t l = 0; l < L; ++l) {
        PersistLayer& P = hl[l];
        P.attn_norm = vec(d); P.wq = mat(d, d); P.wk = mat(d, d); P.wv = mat(d, d); P.wo = mat(d, d); P.ffn_norm = vec(d); P.w1 = mat(F, d); P.w3 = mat(F, d); P.w2 = mat(d, F);
        P.kc = kc + (size_t)l * ctx * d; P.vc = vc + (size_t)l * ctx * d;
    }
    const float* norm = vec(d); const float* output = mat(V, d);
    PersistLayer* dl; CK(hipMalloc(&dl, sizeof(PersistLayer) * L)); CK(hipMemcpy(dl, hl.data(), sizeof(PersistLayer) * L, hipMemcpyHostToDevice));
    auto uc = [&](size_t bytes) { void* p; CK(hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached)); CK(hipMemset(p, 0, bytes)); return p; };
    PersistDecodeArgs a = {};
    a.layers = dl; a.n_layers = L; a.d = d; a.F = F; a.V = V; a.H = H; a.hd = hd;
    a.xa = (float*)uc(d * 4); a.xb = (float*)uc(d * 4); a.q = (float*)uc(d * 4); a.attn = (float*)uc(d * 4); a.g = (float*)uc(F * 4);
    a.norm = norm; a.output = output; CK(hipMalloc(&a.logits, V * 4));
    std::vector<double2> hr((size_t)ctx * hd / 2); for (size_t i = 0; i < hr.size(); ++i) { hr[i].x = cos(0.001 * i); hr[i].y = sin(0.001 * i); }
    double2* rope; CK(hipMalloc(&rope, hr.size() * sizeof(double2))); CK(hipMemcpy(rope, hr.data(), hr.size() * sizeof(double2), hipMemcpyHostToDevice)); a.rope = rope;
    StepParams hsp = {1, past, 0, 0}; StepParams* sp; CK(hipMalloc(&sp, sizeof hsp)); CK(hipMemcpy(sp, &hsp, sizeof hsp, hipMemcpyHostToDevice)); a.sp = sp;
    a.scale = (float)(1.0 / sqrt((double)hd));
    a.ctl.count = (unsigned long long*)uc(64); a.ctl.err = (uint32_t*)uc(64); a.ctl.arrivals_per_launch = persist_decode_arrivals(L, nCU, H); a.ctl.timeout_ticks = 2000000;
    float* dummy; CK(hipMalloc(&dummy, F * 4 + 4096)); CK(hipMemset(dummy, 0, F * 4 + 4096)); a.ctl.dummy = dummy;
    CK(hipMalloc(&a.ctl.trace, 3 * PERSIST_TRACE_MAX * 8)); CK(hipMemset(a.ctl.trace, 0, 3 * PERSIST_TRACE_MAX * 8));
    float* x0; CK(hipMalloc(&x0, d * 4)); hipLaunchKernelGGL(k_fill, dim3(16), dim3(256), 0, st, x0, (size_t)d, 777u, 1.0f, 0.f);
    CK(hipStreamSynchronize(st));
    auto k = k_decode_persist<2, 6>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS_BYTES));
    float best = 1e30f;
    for (int r = 0; r < 6; ++r) {
        CK(hipMemcpyAsync(a.xa, x0, d * 4, hipMemcpyDeviceToDevice, st));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(k, dim3(nCU), dim3(PTH), P_LDS_BYTES, st, a);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (r) best = ms < best ? ms : best;
    }
    uint32_t err; CK(hipMemcpy(&err, a.ctl.err, 4, hipMemcpyDeviceToHost));
    const double bytes = ((double)per_layer * L + (double)V * d + d) * 4;
    printf("k_decode_persist<2,6>  layers %u past %u: %.1f us per launch, %.1f us per layer incl. lm_head share, %.1f GB/s  err=%u\n", L, past, best * 1e3, best * 1e3 / L, bytes / best / 1e6, err);
    std::vector<unsigned long long> tr(3 * PERSIST_TRACE_MAX); CK(hipMemcpy(tr.data(), a.ctl.trace, tr.size() * 8, hipMemcpyDeviceToHost));
    const char* names[10] = {"qkv", "barrier1(+park wo)", "attention+barrier2", "wo", "barrier3(+park w1w3)", "w1w3", "barrier4(+park w2)", "w2", "barrier5(+park next)", "-"};
    const char* wgn[3] = {"wg 0 (head)", "wg mid", "wg last"};
    for (int w = 0; w < 3; ++w) {
        double sum[10] = {0}; const unsigned long long* t = tr.data() + (size_t)w * PERSIST_TRACE_MAX;
        for (uint32_t l = 1; l < L; ++l)   // skip layer 0 (cold start)
            for (int i = 0; i < 9; ++i) sum[i] += (double)(t[l * 10 + i + 1] - t[l * 10 + i]) * 0.01;
        printf("  %-12s", wgn[w]); double tot = 0;
        for (int i = 0; i < 9; ++i) { printf(" %s %.2f |", names[i], sum[i] / (L - 1)); tot += sum[i] / (L - 1); }
        printf(" layer %.2f us; lm_head %.2f us\n", tot, (double)(t[L * 10] - t[L * 10 - 1]) * 0.01);
    }
    return 0;
}
