// tools/gemm_b9_probe.hip — k_gemm_b9 (csrc/kernels_gemm_b9.h: fp32 GEMM as nine exact bf16 MFMA products) against an f64 host product (small
// shapes) and against k_gemm_glds (fp32-input MFMA) on the 13B prefill shapes, HIP-event timed.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Illama.go_amd/csrc -Iinclude -o tools/gemm_b9_probe tools/gemm_b9_probe.hip
#include "kernels_gemm_b9.h"
#include "kernels_stream_q8b.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <math.h>
#include <string.h>
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static void split_rows(const float* x, uint16_t* xs, uint32_t N, uint32_t K) {
    Split3Args sa = {x, xs, (uint64_t)N * K, K, K, K};
    hipLaunchKernelGGL(k_split3_rows, dim3(N), dim3(256), 0, 0, sa);
}
#ifdef B9_PRESPLIT
template <int WN, int WM, int TN, int TM, int NST = B9_GST, bool WPRE = false>
static float time_b9(GemmArgs a, int reps) {
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    size_t lds = NST * gemm_b9_stage_bytes(BN, BM, WPRE);
    if (lds < 82 * 1024) lds = 82 * 1024;
    auto kern = k_gemm_b9<WN, WM, TN, TM, NST, WPRE>;
#else
template <int WN, int WM, int TN, int TM, int NST = B9_GST>
static float time_b9(GemmArgs a, int reps) {
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    size_t lds = NST * gemm_b9_stage_bytes(BN, BM);
    if (lds < 82 * 1024) lds = 82 * 1024;
    auto kern = k_gemm_b9<WN, WM, TN, TM, NST>;
#endif
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM) * a.groups, grid = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WN * WM * 64), lds, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(WN * WM * 64), lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
template <int WN, int WM, int TN, int TM>
static float time_glds(GemmArgs a, int reps) {
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    size_t lds = (size_t)GST * (BN + BM) * 32 * sizeof(float);
    if (lds < 82 * 1024) lds = 82 * 1024;
    auto kern = k_gemm_glds<WN, WM, TN, TM>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const uint32_t tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM) * a.groups, grid = tiles < 256 ? tiles : 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}
static int check(uint32_t N, uint32_t M, uint32_t K) {
    std::vector<float> hx((size_t)N * K), hw((size_t)M * K), y((size_t)N * M), y2((size_t)N * M);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 23); };
    for (auto& v : hx) { const float r = rnd(); v = r * r * r * 3.f; }
    for (auto& v : hw) v = rnd() * 0.05f;
    float *x, *w, *dy, *dy2; uint16_t* xs;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&dy, y.size() * 4)); CK(hipMalloc(&dy2, y.size() * 4)); CK(hipMalloc(&xs, hx.size() * 6));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dy, 0xff, y.size() * 4)); CK(hipMemset(dy2, 0xff, y.size() * 4));
    split_rows(x, xs, N, K);
    GemmArgs a = {};
    a.x = x; a.xs = xs; a.xs_plane = (uint64_t)N * K; a.ldxs = K; a.w[0] = w; a.y[0] = dy; a.groups = 1; a.N = N; a.M = M; a.K = K; a.ldx = K; a.ldy = M;
    const int cfg = getenv("B9_8WAVES") ? atoi(getenv("B9_8WAVES")) : 0;
    if (cfg == 3) time_b9<1, 8, 4, 1, 2>(a, 1); else if (cfg) time_b9<2, 4, 2, 1>(a, 1); else time_b9<2, 2, 2, 2>(a, 1);
#ifdef B9_PRESPLIT   // needs tools/gemm_b9_presplit.patch applied to csrc/ (round 6: measured, not adopted - profiles/r06_gemm_b9_presplit.txt)
    {   // the pre-split-weights build must give the SAME bits (same products, same order)
        uint16_t* wps; float* dy3; CK(hipMalloc(&wps, hw.size() * 6)); CK(hipMalloc(&dy3, y.size() * 4)); CK(hipMemset(dy3, 0xff, y.size() * 4));
        split_rows(w, wps, M, K);
        GemmArgs b = a; b.wp[0] = wps; b.wp_plane = (uint64_t)M * K; b.y[0] = dy3;
        time_b9<1, 8, 4, 1, 2, true>(b, 1);
        a.y[0] = dy; time_b9<1, 8, 4, 1, 2>(a, 1);
        std::vector<float> y3(y.size()), y1(y.size());
        CK(hipMemcpy(y3.data(), dy3, y.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y1.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
        size_t diff = 0; for (size_t i = 0; i < y.size(); ++i) diff += memcmp(&y3[i], &y1[i], 4) != 0;
        printf("   pre-split weights vs split on the fly (128 x 256 tiles): %zu of %zu results differ in their bits\n", diff, y.size());
        hipFree(wps); hipFree(dy3);
        if (cfg != 3) time_b9<2, 2, 2, 2>(a, 1);
    }
#endif
    a.y[0] = dy2;
    time_glds<2, 2, 2, 2>(a, 1);
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y2.data(), dy2, y.size() * 4, hipMemcpyDeviceToHost));
    double e9 = 0, e32 = 0, mx = 0, q9 = 0, q32 = 0; size_t cnt = 0;
    for (uint32_t n = 0; n < N; n += (N > 256 ? 7 : 1)) for (uint32_t m = 0; m < M; m += (M > 512 ? 5 : 1)) {
        double r = 0; for (uint32_t k = 0; k < K; ++k) r += (double)hx[(size_t)n * K + k] * hw[(size_t)m * K + k];
        double a9 = fabs(r - y[(size_t)n * M + m]), a32 = fabs(r - y2[(size_t)n * M + m]);
        if (!(a9 == a9)) a9 = 1e30;
        e9 = fmax(e9, a9); e32 = fmax(e32, a32); mx = fmax(mx, fabs(r)); q9 += a9 * a9; q32 += a32 * a32; ++cnt;
    }
#ifndef B9_PRODUCTS
#define B9_PRODUCTS_SHOWN 8
#else
#define B9_PRODUCTS_SHOWN B9_PRODUCTS
#endif
    printf("check N=%u M=%u K=%u vs the f64 product (max|y| %.3g): bf16 x %d max err %.4e rms %.4e | fp32 MFMA max err %.4e rms %.4e  %s\n", N, M, K, mx, B9_PRODUCTS_SHOWN, e9, sqrt(q9 / cnt), e32, sqrt(q32 / cnt),
           e9 <= 1.5 * e32 + 1e-7 * mx ? "ok" : "MISMATCH");
    hipFree(x); hipFree(w); hipFree(dy); hipFree(dy2); hipFree(xs);
    return e9 <= 1.5 * e32 + 1e-7 * mx ? 0 : 1;
}
int main() {
    int bad = 0;
    bad |= check(128, 128, 256); bad |= check(200, 352, 1024); bad |= check(1024, 640, 5120);
    // 13B prefill, N = 1024
    const uint32_t N = 1024, d = 5120, F = 13824;
    float *x, *w, *y; uint16_t* xs;
    const size_t maxw = (size_t)2 * F * d;
    CK(hipMalloc(&x, (size_t)N * F * 4)); CK(hipMalloc(&w, maxw * 4)); CK(hipMalloc(&y, (size_t)N * 3 * F * 4)); CK(hipMalloc(&xs, (size_t)N * F * 6));
    {   // random fill on the host once (small pattern repeated)
        std::vector<float> h(1 << 22); uint32_t s = 7; for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 23) * 0.05f; }
        for (size_t o = 0; o < maxw; o += h.size()) CK(hipMemcpy(w + o, h.data(), std::min(h.size(), maxw - o) * 4, hipMemcpyHostToDevice));
        for (size_t o = 0; o < (size_t)N * F; o += h.size()) CK(hipMemcpy(x + o, h.data(), std::min(h.size(), (size_t)N * F - o) * 4, hipMemcpyHostToDevice));
    }
    struct Sh { const char* name; uint32_t M, K, groups; } shapes[] = {{"wq|wk|wv", d, d, 3}, {"wo", d, d, 1}, {"w1|w3", F, d, 2}, {"w2", d, F, 1}};
    for (const Sh& sh : shapes) {
        split_rows(x, xs, N, sh.K);
        GemmArgs a = {};
        a.x = x; a.xs = xs; a.xs_plane = (uint64_t)N * sh.K; a.ldxs = sh.K; a.groups = sh.groups; a.N = N; a.M = sh.M; a.K = sh.K; a.ldx = sh.K; a.ldy = sh.M;
        for (uint32_t g = 0; g < sh.groups; ++g) { a.w[g] = w + (size_t)g * sh.M * sh.K; a.y[g] = y + (size_t)g * N * sh.M; }
        const double fl = 2.0 * N * sh.M * sh.K * sh.groups;
        const int cfg = getenv("B9_8WAVES") ? atoi(getenv("B9_8WAVES")) : 0;
#ifdef B9_TRACE
        unsigned long long* clk; CK(hipMalloc(&clk, 16)); CK(hipMemset(clk, 0, 16));
        a.clk = clk;
#endif
        const float t9 = cfg == 3 ? time_b9<1, 8, 4, 1, 2>(a, 5) : cfg == 1 ? time_b9<2, 4, 2, 1>(a, 5) : time_b9<2, 2, 2, 2>(a, 5);
#ifdef B9_TRACE
        unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
        printf("    shader clock while k_gemm_b9 runs: %.3f GHz (%llu clocks in %.1f us)\n", (double)hc[0] / ((double)hc[1] * 10.0), hc[0], (double)hc[1] / 100.0);
        a.clk = nullptr; hipFree(clk);
#endif
#ifdef B9_PRESPLIT
        {   // the same launch on pre-split weights
            uint16_t* wps; CK(hipMalloc(&wps, (size_t)sh.groups * sh.M * sh.K * 6));
            split_rows(w, wps, sh.groups * sh.M, sh.K);
            GemmArgs b = a; b.wp_plane = (uint64_t)sh.groups * sh.M * sh.K;
            for (uint32_t g = 0; g < sh.groups; ++g) b.wp[g] = wps + (size_t)g * sh.M * sh.K;
            const float tp = time_b9<1, 8, 4, 1, 2, true>(b, 5);
#ifdef B9_TRACE
            unsigned long long hc2[2]; CK(hipMalloc(&b.clk, 16)); CK(hipMemset(b.clk, 0, 16)); time_b9<1, 8, 4, 1, 2, true>(b, 1); CK(hipMemcpy(hc2, b.clk, 16, hipMemcpyDeviceToHost)); hipFree(b.clk);
            printf("    pre-split weights: %8.1f us = %6.1f TFLOP/s, shader clock %.3f GHz\n", tp * 1e3, fl / tp / 1e9, (double)hc2[0] / ((double)hc2[1] * 10.0));
#else
            printf("    pre-split weights: %8.1f us = %6.1f TFLOP/s\n", tp * 1e3, fl / tp / 1e9);
#endif
            hipFree(wps);
        }
#endif
        const float t32 = time_glds<2, 2, 2, 2>(a, 5), t160 = time_glds<4, 1, 1, 5>(a, 5);
        printf("          fp32 MFMA, 128 x 160 tiles: %8.1f us\n", t160 * 1e3);
        printf("%-9s N=%u M=%u x %u K=%u: bf16 x 9 %8.1f us = %6.1f TFLOP/s (%.2f of the fp32 MFMA peak) | fp32 MFMA (k_gemm_glds 128x128) %8.1f us = %6.1f TFLOP/s\n", sh.name, N, sh.M, sh.groups, sh.K,
               t9 * 1e3, fl / t9 / 1e9, fl / t9 / 1e9 / 157.3, t32 * 1e3, fl / t32 / 1e9);
    }
    return bad;
}
