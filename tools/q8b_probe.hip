// tools/q8b_probe.hip — k_stream_q8b (csrc/kernels_stream_q8b.h: block-int8 weights on the bf16 matrix pipe through the lossless
// three-piece split of the activations) against a double-precision host product (round 4's k_stream_q8, fp32-input MFMA, was timed beside it
// until it was removed: profiles/r05_q8b_probe.txt keeps those numbers).
// usage: q8b_probe M K N [KC [groups [epi [ksplit]]]]      (7B: w1|w3 = 11008 4096 n 256 2 1, wq|wk|wv = 4096 4096 n 256 3, wo = 4096 4096 n, w2 = 4096 11008 n)
// Timing rotates over enough copies of the weights to exceed the 256 MB Infinity Cache (a re-read matrix would come out of it).
#define Q8B_TRACE
#include "../llama.go_amd/csrc/kernels_stream_q8b.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
#ifndef Q8B_PROBE_TH
#define Q8B_PROBE_TH 1024
#endif
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static int g_kc = 256, g_nimg = 0;
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
template <int MAXT, int NCT, int KC, int NIMG, int XR> static void run_b(const Copies& c, int nCU, double wbytes) {
    if constexpr (NIMG >= 2) {
    const size_t lds = (size_t)NIMG * stream_q8b_image_bytes(MAXT, XR, KC);
    constexpr int TH = (KC == 256 && Q8B_PROBE_TH == 512) ? 512 : 1024;
    auto kern = k_stream_q8b<MAXT, NCT, KC, NIMG, XR, TH>;
    const size_t req = std::max<size_t>(lds, 82 * 1024);
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)req));
    const uint32_t S = c.a[0].ksplit > 1 ? c.a[0].ksplit : 1;
    const double us = time_us([&](int i) { hipLaunchKernelGGL(kern, dim3(nCU / S * S), dim3(TH), req, 0, c.a[i]); }, (int)c.a.size());
    printf("%d threads: ", TH);
    printf("k_stream_q8b<%d,%d,%d,%d,%d>%s (bf16 x 3): %.2f us per launch, %.1f GB/s of weight bytes (LDS %zu B)\n", MAXT, NCT, KC, NIMG, XR, S > 1 ? " K-split" : "", us, wbytes / us / 1e3, lds);
#ifdef Q8B_TRACE
    { unsigned long long tr[16]; CK(hipMemcpy(tr, c.a[0].trace, sizeof tr, hipMemcpyDeviceToHost));
      for (int w = 0; w < 2; ++w) { const int b = w * 8; auto us_ = [&](int i) { return (double)(tr[b + i] - tr[b]) / 100.0; };
        printf("   wave %2d (us from its start): first barrier passed %.2f | loop end %.2f | epilogue start %.2f end %.2f | in the loop: waiting for its DMAs %.2f, at barriers + issuing %.2f, computing %.2f\n",
               w ? 15 : 0, us_(2), us_(3), us_(6), us_(7), tr[b + 4] / 100.0, tr[b + 5] / 100.0, tr[b + 1] / 100.0); } }
#endif
    } else printf("k_stream_q8b<%d,%d,%d>: images do not fit\n", MAXT, NCT, KC);
}
constexpr int q8b_nimg(int maxt, int xr, int kc, int want) { int n = (int)(160 * 1024 / stream_q8b_image_bytes(maxt, xr, kc)); n = n < 4 ? n : 4; return want && want < n ? want : n; }
template <int MAXT, int NCT, int KC, int XR> static void run_bk(const Copies& c, int nCU, double wbytes) {
    if (g_nimg == 2) run_b<MAXT, NCT, KC, q8b_nimg(MAXT, XR, KC, 2), XR>(c, nCU, wbytes);
    else if (g_nimg == 3) run_b<MAXT, NCT, KC, q8b_nimg(MAXT, XR, KC, 3), XR>(c, nCU, wbytes);
    else run_b<MAXT, NCT, KC, q8b_nimg(MAXT, XR, KC, 0), XR>(c, nCU, wbytes);
}
template <int MAXT, int NCT, int XR> static void run(const Copies& c, const Copies&, int nCU, double wbytes) {
    if (g_kc == 512) { if constexpr (NCT <= 2 && MAXT <= 4) run_bk<MAXT, NCT, 512, XR>(c, nCU, wbytes); else printf("KC 512: up to four tiles and two column tiles\n"); }
    else if (g_kc == 256) { if constexpr (NCT <= 2) run_bk<MAXT, NCT, 256, XR>(c, nCU, wbytes); else run_bk<MAXT, NCT, 128, XR>(c, nCU, wbytes); }
    else run_bk<MAXT, NCT, 128, XR>(c, nCU, wbytes);
}
int main(int argc, char** argv) {
    const uint32_t M = argc > 1 ? atoi(argv[1]) : 256, K = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 8;
    if (argc > 4) g_kc = atoi(argv[4]);
    const uint32_t G = argc > 5 ? atoi(argv[5]) : 1, epi = argc > 6 ? atoi(argv[6]) : 0, KS = argc > 7 ? atoi(argv[7]) : 1;   // groups (matrices of M rows each), epilogue (1 = silu*mul over groups 0,1)
    if (getenv("Q8B_IMAGES")) g_nimg = atoi(getenv("Q8B_IMAGES"));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); const int nCU = p.multiProcessorCount;
    const size_t MW = (size_t)M * G;
    std::vector<float> W(MW * K), X((size_t)N * K);
    unsigned s = 1; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); };
    for (auto& v : W) v = rnd() * 0.02f;
    for (auto& v : X) { const float r = rnd(); v = r * r * r * 4.0f; }   // a wide spread of magnitudes
    std::vector<signed char> Q(MW * K); std::vector<float> Sc(MW * K / 32);
    for (size_t b = 0; b < Sc.size(); ++b) {
        float mx = 0; for (int i = 0; i < 32; ++i) mx = std::max(mx, fabsf(W[b * 32 + i]));
        const float dd = mx / 127.0f; Sc[b] = dd;
        for (int i = 0; i < 32; ++i) { float t = dd > 0 ? rintf(W[b * 32 + i] / dd) : 0.f; t = std::min(std::max(t, -127.f), 127.f); Q[b * 32 + i] = (signed char)t; W[b * 32 + i] = dd * (float)(int)t; }
    }
    const double wbytes = (double)MW * K * 36.0 / 32;
    const int ncopies = getenv("Q8B_COPIES") ? atoi(getenv("Q8B_COPIES")) : (int)std::max(1.0, std::ceil(600e6 / wbytes));
    const uint32_t NP = (N + 15) / 16 * 16;
    float* dX; uint16_t* dXs; CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dXs, (size_t)3 * NP * K * 2)); CK(hipMemset(dXs, 0, (size_t)3 * NP * K * 2));
    CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    Split3Args sa = {dX, dXs, (uint64_t)NP * K, K, K, K};
    hipLaunchKernelGGL(k_split3_rows, dim3(N), dim3(256), 0, 0, sa); CK(hipDeviceSynchronize());
    {   // the split is exact: hi + mid + lo == x bit for bit
        std::vector<uint16_t> xs((size_t)3 * NP * K); CK(hipMemcpy(xs.data(), dXs, xs.size() * 2, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t c = 0; c < N; ++c) for (size_t k = 0; k < K; ++k) {
            double sum = 0; for (int pl = 0; pl < 3; ++pl) { const uint32_t b = (uint32_t)xs[(size_t)pl * NP * K + c * K + k] << 16; float f; memcpy(&f, &b, 4); sum += (double)f; }
            if (sum != (double)X[c * K + k]) ++bad;
        }
        printf("split3: %zu of %zu activations differ from hi + mid + lo\n", bad, (size_t)N * K);
    }
    Copies cb, ca;
    float* dY0 = nullptr; float* dY1 = nullptr;
    for (int i = 0; i < ncopies; ++i) {
        signed char* dQ; float* dS; CK(hipMalloc(&dQ, Q.size())); CK(hipMalloc(&dS, Sc.size() * 4));
        CK(hipMemcpy(dQ, Q.data(), Q.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dS, Sc.data(), Sc.size() * 4, hipMemcpyHostToDevice));
        if (i == 0) { CK(hipMalloc(&dY0, (size_t)N * MW * 4)); CK(hipMalloc(&dY1, (size_t)N * MW * 4)); CK(hipMemset(dY0, 0xFF, (size_t)N * MW * 4)); CK(hipMemset(dY1, 0xFF, (size_t)N * MW * 4)); }
        StreamArgs a = {};
        for (uint32_t g = 0; g < G; ++g) { a.w[g] = (const float*)(dQ + (size_t)g * M * K); a.ws[g] = dS + (size_t)g * M * (K / 32); a.y[g] = dY0 + (size_t)g * N * M; }
        a.x = dX; a.groups = G; a.M = M; a.K = K; a.n = N; a.ldx = K; a.ldy = M; a.epi = epi;
        a.xs = dXs; a.xs_plane = (uint64_t)NP * K; a.ldxs = K;
#ifdef Q8B_TRACE
        if (i == 0) { CK(hipMalloc(&a.trace, 256)); CK(hipMemset(a.trace, 0, 256)); } else a.trace = nullptr;
#endif
        cb.a.push_back(a);
        for (uint32_t g = 0; g < G; ++g) a.y[g] = dY1 + (size_t)g * N * M;
        ca.a.push_back(a);
    }
    float* dP = nullptr;
    if (KS > 1) {   // groups of KS workgroups split the contraction; partial sums [KS][N][M], added on the host for the check
        CK(hipMalloc(&dP, (size_t)KS * N * MW * 4)); CK(hipMemset(dP, 0xFF, (size_t)KS * N * MW * 4));
        for (auto& a : cb.a) { a.ksplit = KS; a.ysplit = (uint64_t)N * MW; a.y[0] = dP; }
    }
    const uint32_t ngrp = nCU / KS;
    const uint32_t T = (epi == 1 ? 2 : 1) * ((M / 16 * (epi == 1 ? 1 : G) + ngrp - 1) / ngrp);   // tiles per workgroup (pairs under silu*mul)
    printf("M %u x %u groups, K %u, N %u: %d weight copies of %.1f MB, <= %u tiles per workgroup\n", M, G, K, N, ncopies, wbytes / 1e6, T);
#define GO(MT) { if (N <= 8) run<MT, 1, 8>(cb, ca, nCU, wbytes); else if (N <= 16) run<MT, 1, 16>(cb, ca, nCU, wbytes); else if (N <= 32) run<MT, 2, 32>(cb, ca, nCU, wbytes); else if (N <= 48) run<MT, 3, 48>(cb, ca, nCU, wbytes); else run<MT, 4, 64>(cb, ca, nCU, wbytes); }
    if (T <= 1) GO(1) else if (T <= 2) GO(2) else if (T <= 3) GO(3) else if (T <= 4) GO(4) else if (T <= 6) GO(6) else if (T <= 8) GO(8) else { printf("more than eight tiles per workgroup\n"); return 1; }
    if (getenv("Q8B_SKIP_CHECK")) return 0;
    std::vector<float> Y0((size_t)N * MW), Y1((size_t)N * MW);
    CK(hipMemcpy(Y0.data(), dY0, Y0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(Y1.data(), dY1, Y1.size() * 4, hipMemcpyDeviceToHost));
    if (KS > 1) {
        std::vector<float> P((size_t)KS * N * MW); CK(hipMemcpy(P.data(), dP, P.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < Y0.size(); ++i) { float t = P[i]; for (uint32_t s2 = 1; s2 < KS; ++s2) t += P[(size_t)s2 * N * MW + i]; Y0[i] = t; }
        Y1 = Y0;
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
    printf("vs the f64 product of the dequantised weights (max |ref| %.3e): max err %.3e rms %.3e\n", scale, worst0, sqrt(rms0 / cnt));
    return (worst0 > 1e-4 * scale) ? 2 : 0;
}
