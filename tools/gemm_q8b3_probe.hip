// tools/gemm_q8b3_probe.hip — k_gemm_q8b3 (csrc/kernels_gemm_b9.h: block-int8 weights x three bf16 planes of X, three MFMAs per block) against an f64
// host product of the dequantised weights (small shapes) and against the dequantising fp32-MFMA k_gemm_q8 on the 13B / 7B prefill shapes, HIP-event timed.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Illama.go_amd/csrc -Iinclude -o tools/gemm_q8b3_probe tools/gemm_q8b3_probe.hip
#include "kernels_gemm_b9.h"
#include "kernels_stream_q8b.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static void split_rows(const float* x, uint16_t* xs, uint32_t N, uint32_t K) {
    Split3Args sa = {x, xs, (uint64_t)N * K, K, K, K};
    hipLaunchKernelGGL(k_split3_rows, dim3(N), dim3(256), 0, 0, sa);
}
static float time_q3(GemmArgs a, int reps) {
    const size_t lds = gemm_q8b3_lds_bytes(4);
    auto kern = k_gemm_q8b3<4>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t tiles = ((a.N + 127) / 128) * ((a.M + 255) / 256) * a.groups, grid = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
static float time_q8(GemmArgs a, int reps) {
    auto kern = k_gemm_q8<2, 2, 2, 2>;
    const size_t lds = std::max<size_t>((size_t)2 * (128 + 128) * 32 * sizeof(float), 82 * 1024);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t tiles = ((a.N + 127) / 128) * ((a.M + 127) / 128) * a.groups, grid = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}
static int check(uint32_t N, uint32_t M, uint32_t K, bool resid) {
    std::vector<float> hx((size_t)N * K), hs((size_t)M * K / 32), hr((size_t)N * M);
    std::vector<int8_t> hq((size_t)M * K);
    uint32_t s = 12345 + N + M + K;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (auto& v : hx) v = (float)((int)rnd() - (1 << 23)) / (float)(1 << 23);
    for (auto& v : hq) v = (int8_t)((int)(rnd() % 255) - 127);
    for (auto& v : hs) v = (0.5f + (float)(rnd() % 1024) / 1024.f) / 127.f / sqrtf((float)K);
    for (auto& v : hr) v = (float)((int)rnd() - (1 << 23)) / (float)(1 << 23);
    float *x, *sc, *y, *y8, *r; int8_t* q; uint16_t* xs;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&sc, hs.size() * 4)); CK(hipMalloc(&q, hq.size())); CK(hipMalloc(&y, (size_t)N * M * 4)); CK(hipMalloc(&y8, (size_t)N * M * 4));
    CK(hipMalloc(&r, (size_t)N * M * 4)); CK(hipMalloc(&xs, hx.size() * 6));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sc, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(q, hq.data(), hq.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(r, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(y, 0xff, (size_t)N * M * 4));
    split_rows(x, xs, N, K);
    GemmArgs a = {};
    a.x = x; a.xs = xs; a.xs_plane = (uint64_t)N * K; a.ldxs = K; a.groups = 1; a.N = N; a.M = M; a.K = K; a.ldx = K; a.ldy = M;
    a.w[0] = (const float*)q; a.ws[0] = sc; a.y[0] = y; a.r[0] = resid ? r : nullptr;
    time_q3(a, 1);
    a.y[0] = y8; time_q8(a, 1);
    std::vector<float> hy((size_t)N * M), hy8((size_t)N * M);
    CK(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy8.data(), y8, hy8.size() * 4, hipMemcpyDeviceToHost));
    double e3 = 0, e8 = 0, mx = 0;
    for (uint32_t n = 0; n < N; ++n)
        for (uint32_t m = 0; m < M; ++m) {
            double ref = resid ? hr[(size_t)n * M + m] : 0.0;
            for (uint32_t k = 0; k < K; ++k) ref += (double)(float)((float)hq[(size_t)m * K + k] * hs[(size_t)m * (K / 32) + k / 32]) * (double)hx[(size_t)n * K + k];
            mx = fmax(mx, fabs(ref));
            e3 = fmax(e3, fabs(ref - (double)hy[(size_t)n * M + m])); e8 = fmax(e8, fabs(ref - (double)hy8[(size_t)n * M + m]));
        }
    const bool ok = e3 <= 1.5 * e8 + 2e-7 * mx && e3 == e3;
    printf("check N=%u M=%u K=%u%s vs the f64 product of the dequantised weights (max|y| %.3g): three-MFMA max err %.3e | k_gemm_q8 max err %.3e  %s\n", N, M, K, resid ? " +resid" : "", mx, e3, e8, ok ? "ok" : "MISMATCH");
    hipFree(x); hipFree(sc); hipFree(q); hipFree(y); hipFree(y8); hipFree(r); hipFree(xs);
    return ok ? 0 : 1;
}
int main() {
    int bad = 0;
    if (!getenv("Q3_NOCHECK")) bad |= check(128, 256, 512, false); if (!getenv("Q3_NOCHECK")) { bad |= check(200, 352, 1024, true); bad |= check(130, 300, 2048, false); bad |= check(1024, 640, 5120, true); }
    const uint32_t N = 1024;
    struct Sh { const char* name; uint32_t M, K, groups; } shapes[] = {{"13B wq|wk|wv", 5120, 5120, 3}, {"13B wo", 5120, 5120, 1}, {"13B w1|w3", 13824, 5120, 2}, {"13B w2", 5120, 13824, 1},
                                                                      {"7B wq|wk|wv", 4096, 4096, 3}, {"7B wo", 4096, 4096, 1}, {"7B w1|w3", 11008, 4096, 2}, {"7B w2", 4096, 11008, 1}};
    const size_t maxw = (size_t)2 * 13824 * 5120;
    float *x, *sc, *y; int8_t* q; uint16_t* xs;
    CK(hipMalloc(&x, (size_t)N * 13824 * 4)); CK(hipMalloc(&q, maxw)); CK(hipMalloc(&sc, maxw / 32 * 4)); CK(hipMalloc(&y, (size_t)N * 3 * 13824 * 4)); CK(hipMalloc(&xs, (size_t)N * 13824 * 6));
    {
        std::vector<float> h(1 << 22); uint32_t s = 7; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 23) * 0.05f; }
        for (size_t o = 0; o < (size_t)N * 13824; o += h.size()) CK(hipMemcpy(x + o, h.data(), std::min(h.size(), (size_t)N * 13824 - o) * 4, hipMemcpyHostToDevice));
        for (size_t o = 0; o < maxw / 32; o += h.size()) CK(hipMemcpy(sc + o, h.data(), std::min(h.size(), maxw / 32 - o) * 4, hipMemcpyHostToDevice));
        for (size_t o = 0; o < maxw; o += h.size() * 4) CK(hipMemcpy(q + o, h.data(), std::min(h.size() * 4, maxw - o), hipMemcpyHostToDevice));
    }
    for (const Sh& sh : shapes) {
        split_rows(x, xs, N, sh.K);
        GemmArgs a = {};
        a.x = x; a.xs = xs; a.xs_plane = (uint64_t)N * sh.K; a.ldxs = sh.K; a.groups = sh.groups; a.N = N; a.M = sh.M; a.K = sh.K; a.ldx = sh.K; a.ldy = sh.M;
        for (uint32_t g = 0; g < sh.groups; ++g) { a.w[g] = (const float*)(q + (size_t)g * sh.M * sh.K); a.ws[g] = sc + (size_t)g * sh.M * sh.K / 32; a.y[g] = y + (size_t)g * N * sh.M; }
        const double fl = 2.0 * N * sh.M * sh.K * sh.groups;
#ifdef B9_TRACE
        unsigned long long* clk; CK(hipMalloc(&clk, 16)); CK(hipMemset(clk, 0, 16));
        a.clk = clk;
#endif
        const float t3 = time_q3(a, 5);
#ifdef B9_TRACE
        unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
        printf("    shader clock while k_gemm_q8b3 runs: %.3f GHz (%llu clocks in %.1f us)\n", (double)hc[0] / ((double)hc[1] * 10.0), hc[0], (double)hc[1] / 100.0);
        a.clk = nullptr; hipFree(clk);
#endif
        const float t8 = getenv("Q3_SKIP_Q8") ? 1.f : time_q8(a, 5);
        printf("%-13s N=%u M=%u x %u K=%u: three-MFMA %8.1f us = %6.1f TFLOP/s | dequantising fp32 MFMA (k_gemm_q8 128x128) %8.1f us = %6.1f TFLOP/s\n", sh.name, N, sh.M, sh.groups, sh.K, t3 * 1e3, fl / t3 / 1e9, t8 * 1e3,
               fl / t8 / 1e9);
    }
    return bad;
}
