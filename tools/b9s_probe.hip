// tools/b9s_probe.hip — k_stream_b9 (csrc/kernels_stream_b9.h: fp32 weights on the bf16 matrix pipe through the lossless 3 x 3 split) against a
// double-precision host product, timed beside k_stream_dma (fp32-input MFMA) on the same matrices.
// usage: b9s_probe M K N [- [groups [epi [ksplit]]]]      (7B: w1|w3 = 11008 4096 n 64 2 1, wq|wk|wv = 4096 4096 n 64 3, wo = 4096 4096 n 64 1 0 4, w2 = 4096 11008 n 64 1 0 4)
// env: B9S_IMAGES (cap of the weight ring's depth), B9S_SKIP_CHECK, B9S_COPIES
// Timing rotates over enough copies of the weights to exceed the 256 MB Infinity Cache (a re-read matrix would come out of it).
#define Q8B_TRACE
#include "../llama.go_amd/csrc/kernels_stream_b9.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#ifndef B9S_TAG
#define B9S_TAG ""
#endif
#ifndef B9S_NPROD
#define B9S_NPROD 9
#endif
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static int g_kc = 2, g_nimg = 0;
struct Copies { std::vector<StreamArgs> a; };
template <typename F> static double time_us(F&& launch, int ncopies) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < ncopies; ++i) launch(i);
    const int reps = 5 * ncopies;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch(i % ncopies);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}
template <int MAXT, int NCT, int NIMG> static void run_b(const Copies& c, int nCU, double wbytes) {
    if constexpr (NIMG >= 2 && NCT <= 4) {
    const size_t lds = (size_t)NIMG * stream_b9_image_bytes(MAXT, NCT);
    auto kern = k_stream_b9<MAXT, NCT, NIMG, B9S_NPROD>;
    const size_t req = std::max<size_t>(lds, 82 * 1024);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)req));
    const uint32_t S = c.a[0].ksplit > 1 ? c.a[0].ksplit : 1;
    const double us = time_us([&](int i) { hipLaunchKernelGGL(kern, dim3(nCU / S * S), dim3(B9S_TH), req, 0, c.a[i]); }, (int)c.a.size());
    printf("k_stream_b9<%d,%d,%d images> %d products" B9S_TAG "%s: %.2f us per launch, %.1f GB/s of weight bytes (LDS %zu B)\n", MAXT, NCT, NIMG, B9S_NPROD, S > 1 ? " K-split" : "", us,
           wbytes / us / 1e3, lds);
    { unsigned long long tr[2]; CK(hipMemcpy(tr, c.a[0].trace, sizeof tr, hipMemcpyDeviceToHost));
      printf("   one workgroup's life: %.1f us, %.0f shader clocks -> %.0f MHz\n", tr[1] / 100.0, (double)tr[0], (double)tr[0] / (tr[1] / 100.0)); }
    } else printf("k_stream_b9<%d,%d>: not built (more than four column tiles, or two images do not fit)\n", MAXT, NCT);
}
template <int MAXT, int NCT> static void run_bk(const Copies& c, int nCU, double wbytes) {
    if (g_nimg == 2) run_b<MAXT, NCT, stream_b9_nimg(MAXT, NCT <= 4 ? NCT : 4, 2)>(c, nCU, wbytes);
    else if (g_nimg == 3) run_b<MAXT, NCT, stream_b9_nimg(MAXT, NCT <= 4 ? NCT : 4, 3)>(c, nCU, wbytes);
    else run_b<MAXT, NCT, stream_b9_nimg(MAXT, NCT <= 4 ? NCT : 4, 4)>(c, nCU, wbytes);
}
// the fp32-MFMA kernel of the product on the same launch (plain fp32 rows as activations)
template <int MAXT, int NCT> static void run_dma(const Copies& c, int nCU, double wbytes) {
    constexpr int N64 = (int)(160 * 1024 / ((size_t)(MAXT + NCT) * 16 * 64 * 4)) < 4 ? (int)(160 * 1024 / ((size_t)(MAXT + NCT) * 16 * 64 * 4)) : 4;
    if constexpr (N64 >= 2 && (NCT < 7 || MAXT <= 6)) {
        constexpr int CS = NCT == 8 ? 2 : 1;
        auto kern = k_stream_dma<MAXT, NCT, 64, N64, false, CS>;
        const size_t req = std::max<size_t>(stream_dma_lds_bytes(MAXT, NCT, 64, N64), 82 * 1024);
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)req));
        const uint32_t S = c.a[0].ksplit > 1 ? c.a[0].ksplit : 1;
        const double us = time_us([&](int i) { hipLaunchKernelGGL(kern, dim3(nCU / S * S), dim3(2 * ST_TH), req, 0, c.a[i]); }, (int)c.a.size());
        printf("k_stream_dma<%d,%d,64,%d> (fp32 MFMA)%s: %.2f us per launch, %.1f GB/s of weight bytes\n", MAXT, NCT, N64, S > 1 ? " K-split" : "", us, wbytes / us / 1e3);
    }
}
template <int MAXT, int NCT> static void run(const Copies& cb, const Copies& ca, int nCU, double wbytes) {
    run_bk<MAXT, NCT>(cb, nCU, wbytes);
    if (!getenv("B9S_NO_DMA")) run_dma<MAXT, NCT>(ca, nCU, wbytes);
}
int main(int argc, char** argv) {
    const uint32_t M = argc > 1 ? atoi(argv[1]) : 256, K = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 32;
    if (argc > 4) g_kc = atoi(argv[4]);
    const uint32_t G = argc > 5 ? atoi(argv[5]) : 1, epi = argc > 6 ? atoi(argv[6]) : 0, KS = argc > 7 ? atoi(argv[7]) : 1;   // groups (matrices of M rows each), epilogue (1 = silu*mul over groups 0,1)
    if (getenv("B9S_IMAGES")) g_nimg = atoi(getenv("B9S_IMAGES"));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); const int nCU = p.multiProcessorCount;
    const size_t MW = (size_t)M * G;
    std::vector<float> W(MW * K), X((size_t)N * K);
    unsigned s = 1; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); };
    for (auto& v : W) { const float r = rnd(); v = r * (0.25f + r * r) * 0.02f; }
    for (auto& v : X) { const float r = rnd(); v = r * r * r * 4.0f; }   // a wide spread of magnitudes
    const double wbytes = (double)MW * K * 4.0;
    const int ncopies = getenv("B9S_COPIES") ? atoi(getenv("B9S_COPIES")) : (int)std::max(1.0, std::ceil(600e6 / wbytes));
    const uint32_t NP = (N + 15) / 16 * 16;
    float* dX; uint16_t* dXs; CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dXs, (size_t)3 * NP * K * 2)); CK(hipMemset(dXs, 0, (size_t)3 * NP * K * 2));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    Split3Args sa = {dX, dXs, (uint64_t)NP * K, K, K, K};
    hipLaunchKernelGGL(k_split3_rows, dim3(N), dim3(256), 0, 0, sa); CK(hipDeviceSynchronize());
    Copies cb, ca;
    float* dY0 = nullptr; float* dY1 = nullptr;
    for (int i = 0; i < ncopies; ++i) {
        float* dW; CK(hipMalloc(&dW, W.size() * 4));
        CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
        if (i == 0) { CK(hipMalloc(&dY0, (size_t)N * MW * 4)); CK(hipMalloc(&dY1, (size_t)N * MW * 4)); CK(hipMemset(dY0, 0xFF, (size_t)N * MW * 4)); CK(hipMemset(dY1, 0xFF, (size_t)N * MW * 4)); }
        StreamArgs a = {};
        for (uint32_t g = 0; g < G; ++g) { a.w[g] = dW + (size_t)g * M * K; a.y[g] = dY0 + (size_t)g * N * M; }
        a.x = dX; a.groups = G; a.M = M; a.K = K; a.n = N; a.ldx = K; a.ldy = M; a.epi = epi;
        a.xs = dXs; a.xs_plane = (uint64_t)NP * K; a.ldxs = K;
        if (i == 0) { CK(hipMalloc(&a.trace, 256)); CK(hipMemset(a.trace, 0, 256)); } else a.trace = nullptr;
        cb.a.push_back(a);
        a.trace = nullptr;
        for (uint32_t g = 0; g < G; ++g) a.y[g] = dY1 + (size_t)g * N * M;
        ca.a.push_back(a);
    }
    float* dP = nullptr; float* dP1 = nullptr;
    if (KS > 1) {   // groups of KS workgroups split the contraction; partial sums [KS][N][M], added on the host for the check
        CK(hipMalloc(&dP, (size_t)KS * N * MW * 4)); CK(hipMemset(dP, 0xFF, (size_t)KS * N * MW * 4));
        CK(hipMalloc(&dP1, (size_t)KS * N * MW * 4)); CK(hipMemset(dP1, 0xFF, (size_t)KS * N * MW * 4));
        for (auto& a : cb.a) { a.ksplit = KS; a.ysplit = (uint64_t)N * MW; a.y[0] = dP; }
        for (auto& a : ca.a) { a.ksplit = KS; a.ysplit = (uint64_t)N * MW; a.y[0] = dP1; }
    }
    const uint32_t ngrp = nCU / KS;
    const uint32_t T = (epi == 1 ? 2 : 1) * ((M / 16 * (epi == 1 ? 1 : G) + ngrp - 1) / ngrp);   // tiles per workgroup (pairs under silu*mul)
    printf("M %u x %u groups, K %u, N %u: %d weight copies of %.1f MB, <= %u tiles per workgroup\n", M, G, K, N, ncopies, wbytes / 1e6, T);
#define GO(MT) { if (N <= 16) run<MT, 1>(cb, ca, nCU, wbytes); else if (N <= 32) run<MT, 2>(cb, ca, nCU, wbytes); else if (N <= 48) run<MT, 3>(cb, ca, nCU, wbytes); else if (N <= 64) run<MT, 4>(cb, ca, nCU, wbytes); \
                 else if (N <= 96) run<MT, 6>(cb, ca, nCU, wbytes); else run<MT, 8>(cb, ca, nCU, wbytes); }
    if (T <= 1) GO(1) else if (T <= 2) GO(2) else if (T <= 3) GO(3) else if (T <= 4) GO(4) else if (T <= 6) GO(6) else if (T <= 8) GO(8) else { printf("more than eight tiles per workgroup\n"); return 1; }
    if (getenv("B9S_SKIP_CHECK")) return 0;
    std::vector<float> Y0((size_t)N * MW), Y1((size_t)N * MW);
    CK(hipMemcpy(Y0.data(), dY0, Y0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Y1.data(), dY1, Y1.size() * 4, hipMemcpyDeviceToHost));
    if (KS > 1) {
        std::vector<float> P((size_t)KS * N * MW);
        CK(hipMemcpy(P.data(), dP, P.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < Y0.size(); ++i) { float t = P[i]; for (uint32_t s2 = 1; s2 < KS; ++s2) t += P[(size_t)s2 * N * MW + i]; Y0[i] = t; }
        CK(hipMemcpy(P.data(), dP1, P.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < Y1.size(); ++i) { float t = P[i]; for (uint32_t s2 = 1; s2 < KS; ++s2) t += P[(size_t)s2 * N * MW + i]; Y1[i] = t; }
    }
    const uint32_t MO = epi == 1 ? M : (uint32_t)MW;   // silu*mul: one output matrix
    double worst0 = 0, worst1 = 0, scale = 0, rms0 = 0, rms1 = 0; size_t cnt = 0;
    const uint32_t rstep = MO > 2048 ? 7 : 1;
    for (uint32_t c = 0; c < N; ++c) for (uint32_t r = 0; r < MO; r += rstep) {
        auto dot = [&](size_t row) { double t = 0; for (uint32_t k = 0; k < K; ++k) t += (double)W[row * K + k] * X[(size_t)c * K + k]; return t; };
        double ref;
        size_t oi;
        if (epi == 1) { const double s1 = dot(r), s3 = dot((size_t)M + r); ref = s1 / (1.0 + exp(-s1)) * s3; oi = (size_t)c * M + r; }
        else { ref = dot(r); const uint32_t g = r / M; oi = (size_t)g * N * M + (size_t)c * M + (r - g * M); }
        double e0 = fabs(ref - Y0[oi]), e1 = fabs(ref - Y1[oi]); if (!(e0 == e0)) e0 = 1e30; if (!(e1 == e1)) e1 = 1e30;
        worst0 = std::max(worst0, e0); worst1 = std::max(worst1, e1); scale = std::max(scale, fabs(ref)); rms0 += e0 * e0; rms1 += e1 * e1; ++cnt;
    }
    printf("vs the f64 product (max |ref| %.3e): k_stream_b9 max err %.3e rms %.3e | k_stream_dma max err %.3e rms %.3e\n", scale, worst0, sqrt(rms0 / cnt), worst1, sqrt(rms1 / cnt));
    return (worst0 > 1e-4 * scale) ? 2 : 0;
}
