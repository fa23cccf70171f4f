// tools/stream_mm_check.hip — k_stream_mm (csrc/kernels_stream.h) against a double-precision host product, with a map of which
// (16-row tile, 16-column tile) blocks are wrong.  usage: stream_mm_check M K N [KC [mode [ksplit]]]
// mode: 0 first variant, 1 chunk-major weight copy, 2 specialised waves, 4 specialised waves with LDS-DMA, 6 sixteen equal waves (k_stream_eq: one column tile; STREAM_EQ_NORM=1 folds an RMSNorm)   (modes 3 / 5, the round-3 / round-4 block-int8 kernels, went with them in round 5: tools/q8b_probe)
// loaders (k_stream_dma; STREAM_DMA_IMAGES=2..4 images in the ring, default 3; STREAM_DMA_PIPE=1 pipelined operands); ksplit S > 1 (mode 2 / 3 / 4): groups of S
// workgroups split the contraction, k_stream_reduce_norm adds the partials (timed alone and with the reduce pass)
// (The -DSTREAM_PROBE / STREAM_TRACE builds of rounds 2-4 - one traffic class taken out of the loop, per-phase shader clocks - went out of the
// product header in round 5; their results are in profiles/r02d_stream_traffic_probe.txt, r03_stream_*.txt, r04_stream_*.txt.)
#include "../llama.go_amd/csrc/kernels_stream.h"
#include "kernels_stream_mm_r1.h"
#include "kernels_stream_eq.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
static int g_kc = 128;
static float* g_yfinal = nullptr;
template <int MAXT, int NCT, int KC> static void run_kc(const StreamArgs& a, int nCU) {
    const size_t lds = std::max<size_t>(stream_lds_bytes(MAXT, NCT, KC), 82 * 1024);
    CK(hipFuncSetAttribute((const void*)k_stream_mm<MAXT, NCT, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_stream_mm<MAXT, NCT, KC>), dim3(nCU), dim3(ST_TH), lds, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k_stream_mm<MAXT, NCT, KC>), dim3(nCU), dim3(ST_TH), lds, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_stream_mm<%d,%d,%d>: %.2f us per launch (same weights every launch: L2 / MALL may help), %.1f GB/s\n", MAXT, NCT, KC, ms * 200, (double)a.M * a.K * 4 / (ms * 200) / 1e3);
}
static int g_wgpcu = 1;   // STREAM_WGPCU=2: two workgroups per CU (grid 2 x #CU, LDS request = the images only; build with a 128-VGPR cap)
template <int MAXT, int NCT, int KC> static void run2_kc(const StreamArgs& a, int nCU0) {
    const int nCU = nCU0 * g_wgpcu;
    const size_t lds = g_wgpcu == 2 ? stream2_lds_bytes(MAXT, NCT, KC) : std::max<size_t>(stream2_lds_bytes(MAXT, NCT, KC), 82 * 1024);
    if (lds > 160 * 1024 || (g_wgpcu == 2 && lds > 80 * 1024)) { printf("k_stream_mm2<%d,%d,%d>: images do not fit (%zu B)\n", MAXT, NCT, KC, lds); return; }
    const bool q8 = false;
    auto kern = k_stream_mm2<MAXT, NCT, KC>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nCU), dim3(2 * ST_TH), lds, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(nCU), dim3(2 * ST_TH), lds, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_stream_mm2<%d,%d,%d%s> (specialised waves%s): %.2f us per launch, %.1f GB/s of weight bytes\n", MAXT, NCT, KC, q8 ? ",int8" : "", a.ksplit > 1 ? ", K-split" : "", ms * 200,
           (double)a.M * a.K * (q8 ? 36.0 / 32 : 4.0) / (ms * 200) / 1e3);
    if (a.ksplit > 1) {
        StreamReduceArgs r = {}; r.part = a.y[0]; r.stride = a.ysplit; r.y = g_yfinal; r.S = a.ksplit; r.d = a.M; r.ldy = a.ldy;
        hipLaunchKernelGGL(k_stream_reduce_norm, dim3(a.n), dim3(256), 0, 0, r);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) { hipLaunchKernelGGL(kern, dim3(nCU), dim3(2 * ST_TH), lds, 0, a); hipLaunchKernelGGL(k_stream_reduce_norm, dim3(a.n), dim3(256), 0, 0, r); }
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("   with the reduce pass: %.2f us per pair of launches\n", ms * 200);
    }
}
static int g_v2 = 0, g_dma = 0, g_nimg = 3, g_pipe = 0;
template <int MAXT, int NCT, int KC, int NIMG, bool PIPE, int CS = 1> static void run_dma_img(const StreamArgs& a, int nCU) {
    const size_t lds = stream_dma_lds_bytes(MAXT, NCT, KC, NIMG);
    if (lds > 160 * 1024) { printf("k_stream_dma<%d,%d,%d,%d>: images do not fit (%zu B)\n", MAXT, NCT, KC, NIMG, lds); return; }
    auto kern = k_stream_dma<MAXT, NCT, KC, NIMG, PIPE, CS>;
    const size_t req = std::max<size_t>(lds, 82 * 1024);   // one workgroup per CU
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)req));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nCU), dim3(2 * ST_TH), req, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(nCU), dim3(2 * ST_TH), req, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_stream_dma<%d,%d,%d> with %d images%s%s: %.2f us per launch, %.1f GB/s of weight bytes\n", MAXT, NCT, KC, NIMG, PIPE ? ", pipelined operands" : "", a.ksplit > 1 ? ", K-split" : "", ms * 200,
           (double)a.M * a.K * 4.0 / (ms * 200) / 1e3);
    if (a.ksplit > 1) {
        StreamReduceArgs r = {}; r.part = a.y[0]; r.stride = a.ysplit; r.y = g_yfinal; r.S = a.ksplit; r.d = a.M; r.ldy = a.ldy;
        hipLaunchKernelGGL(k_stream_reduce_norm, dim3(a.n), dim3(256), 0, 0, r);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 5; ++i) { hipLaunchKernelGGL(kern, dim3(nCU), dim3(2 * ST_TH), req, 0, a); hipLaunchKernelGGL(k_stream_reduce_norm, dim3(a.n), dim3(256), 0, 0, r); }
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("   with the reduce pass: %.2f us per pair of launches\n", ms * 200);
    }
}
template <int MAXT, int NCT, int KC, int NIMG> static void run_dma_p(const StreamArgs& a, int nCU) {
    if constexpr (NCT == 8) {   // eight column tiles: 2 K-groups x 2 column halves, 64-column chunks, pipelined
        if constexpr (KC == 64 && MAXT <= 6) { if (g_pipe) run_dma_img<MAXT, NCT, KC, NIMG, true, 2>(a, nCU); else run_dma_img<MAXT, NCT, KC, NIMG, false, 2>(a, nCU); }
        else printf("k_stream_dma: eight column tiles run 64-column chunks, up to six row tiles\n");
    } else if constexpr (NCT == 7) {
        if constexpr (KC == 64 && MAXT <= 6) run_dma_img<MAXT, NCT, KC, NIMG, false>(a, nCU);
    } else if constexpr (NCT >= 2) {
        constexpr bool POK = MAXT * NCT * 4 + 2 * (MAXT + NCT) * 4 <= 210;   // (plan.hip: dma_pipe_ok)
        if constexpr (POK) { if (g_pipe) { run_dma_img<MAXT, NCT, KC, NIMG, true>(a, nCU); return; } }
        else if (g_pipe) { printf("k_stream_dma<%d,%d,%d>: two operand sets do not fit the registers\n", MAXT, NCT, KC); return; }
        run_dma_img<MAXT, NCT, KC, NIMG, false>(a, nCU);
    } else printf("k_stream_dma: two column tiles on\n");
}
template <int MAXT, int NCT, int KC> static void run_dma(const StreamArgs& a, int nCU) {
    if (g_nimg == 2) run_dma_p<MAXT, NCT, KC, 2>(a, nCU); else if (g_nimg == 4) run_dma_p<MAXT, NCT, KC, 4>(a, nCU); else if (g_nimg == 5) run_dma_p<MAXT, NCT, KC, 5>(a, nCU); else run_dma_p<MAXT, NCT, KC, 3>(a, nCU);
}
static int g_eq = 0;
template <int MAXT, int NIMG> static void run_eq_img(const StreamArgs& a, int nCU) {
    const size_t lds = std::max<size_t>(stream_eq_lds_bytes(MAXT, NIMG), 82 * 1024);
    if (stream_eq_lds_bytes(MAXT, NIMG) > 160 * 1024) { printf("k_stream_eq<%d,%d>: images do not fit\n", MAXT, NIMG); return; }
    auto kern = k_stream_eq<MAXT, NIMG>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(nCU), dim3(SEQ_TH), lds, 0, a);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(nCU), dim3(SEQ_TH), lds, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_stream_eq<%d> with %d images%s: %.2f us per launch, %.1f GB/s of weight bytes\n", MAXT, NIMG, a.gamma ? ", folded norm" : "", ms * 200, (double)a.M * a.K * 4.0 / (ms * 200) / 1e3);
}
template <int MAXT> static void run_eq(const StreamArgs& a, int nCU) {
    if (g_nimg == 2) run_eq_img<MAXT, 2>(a, nCU); else if (g_nimg == 3) run_eq_img<MAXT, 3>(a, nCU); else if (g_nimg == 5) run_eq_img<MAXT, 5>(a, nCU); else run_eq_img<MAXT, 4>(a, nCU);
}
template <int MAXT, int NCT> static void run(const StreamArgs& a, int nCU) {
    if (g_eq) { if constexpr (NCT == 1) run_eq<MAXT>(a, nCU); else printf("k_stream_eq: one column tile\n"); return; }
    if constexpr (NCT >= 7) { run_dma<MAXT, NCT, 64>(a, nCU); return; }   // (mode 4 only: the other kernels are built for up to six column tiles)
    else {
    if (g_dma) { if (g_kc == 64) run_dma<MAXT, NCT, 64>(a, nCU); else run_dma<MAXT, NCT, 128>(a, nCU); return; }
    if (g_v2) { if (g_kc >= 256) run2_kc<MAXT, NCT, 256>(a, nCU); else if (g_kc == 64) run2_kc<MAXT, NCT, 64>(a, nCU); else run2_kc<MAXT, NCT, 128>(a, nCU); return; }
    constexpr int R = MAXT + NCT;
    if (g_kc == 512 && R * 512 <= 2048) run_kc<MAXT, NCT, (R * 512 <= 2048 ? 512 : 128)>(a, nCU);
    else if (g_kc >= 256 && R * 256 <= 2048) run_kc<MAXT, NCT, (R * 256 <= 2048 ? 256 : 128)>(a, nCU);
    else run_kc<MAXT, NCT, 128>(a, nCU);
    }
}
int main(int argc, char** argv) {
    const uint32_t M = argc > 1 ? atoi(argv[1]) : 256, K = argc > 2 ? atoi(argv[2]) : 1024, N = argc > 3 ? atoi(argv[3]) : 33;
    if (argc > 4) g_kc = atoi(argv[4]);
    const bool tiled = argc > 5 && atoi(argv[5]) == 1;
    g_v2 = argc > 5 && (atoi(argv[5]) == 2 || atoi(argv[5]) == 3 || atoi(argv[5]) == 4);   // (mode 4 takes mode 2's dispatch over the tile counts)
    g_dma = argc > 5 && atoi(argv[5]) == 4;
    g_eq = argc > 5 && atoi(argv[5]) == 6;
    if (g_eq) g_v2 = 1;
    if (getenv("STREAM_DMA_IMAGES")) g_nimg = atoi(getenv("STREAM_DMA_IMAGES"));
    if (getenv("STREAM_DMA_PIPE")) g_pipe = atoi(getenv("STREAM_DMA_PIPE"));
    const bool q8 = false;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); const int nCU = p.multiProcessorCount;
    std::vector<float> W((size_t)M * K), X((size_t)N * K), Y((size_t)N * M);
    unsigned s = 1; auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); };
    for (auto& v : W) v = rnd(); for (auto& v : X) v = rnd();
    float *dW, *dX, *dY; CK(hipMalloc(&dW, W.size() * 4)); CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dY, Y.size() * 4));
    if (tiled) {   // chunk-major copy: [K / KC][M][KC]
        std::vector<float> Wt_(W.size()); const uint32_t KCc = (uint32_t)g_kc;
        for (uint32_t r = 0; r < M; ++r) for (uint32_t k = 0; k < K; ++k) Wt_[((size_t)(k / KCc) * M + r) * KCc + k % KCc] = W[(size_t)r * K + k];
        CK(hipMemcpy(dW, Wt_.data(), W.size() * 4, hipMemcpyHostToDevice));
    } else
    CK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice)); CK(hipMemset(dY, 0xFF, Y.size() * 4));
    float* dS = nullptr;
    if (q8) {   // block-int8 planes of the same matrix; the reference product then uses the dequantised values
        std::vector<signed char> Q((size_t)M * K); std::vector<float> S((size_t)M * K / 32);
        for (size_t b = 0; b < S.size(); ++b) {
            float mx = 0; for (int i = 0; i < 32; ++i) mx = std::max(mx, fabsf(W[b * 32 + i]));
            const float dd = mx / 127.0f; S[b] = dd;
            for (int i = 0; i < 32; ++i) { float t = dd > 0 ? rintf(W[b * 32 + i] / dd) : 0.f; t = std::min(std::max(t, -127.f), 127.f); Q[b * 32 + i] = (signed char)t; W[b * 32 + i] = dd * (float)(int)t; }
        }
        CK(hipMemcpy(dW, Q.data(), Q.size(), hipMemcpyHostToDevice));
        CK(hipMalloc(&dS, S.size() * 4)); CK(hipMemcpy(dS, S.data(), S.size() * 4, hipMemcpyHostToDevice));
    }
    float* dG = nullptr;
    std::vector<float> Gm(K, 1.0f);
    if (getenv("STREAM_EQ_NORM")) {   // fold an RMSNorm: the checked product is W . (gamma * x * scale(x))
        for (auto& v : Gm) v = 1.0f + 0.1f * rnd();
        CK(hipMalloc(&dG, K * 4)); CK(hipMemcpy(dG, Gm.data(), K * 4, hipMemcpyHostToDevice));
        for (uint32_t c = 0; c < N; ++c) {
            double ss = 0; for (uint32_t k = 0; k < K; ++k) ss += (double)X[(size_t)c * K + k] * X[(size_t)c * K + k];
            const float sc = (float)(1.0 / sqrt(ss / K + 1e-5));
            for (uint32_t k = 0; k < K; ++k) X[(size_t)c * K + k] = Gm[k] * (X[(size_t)c * K + k] * sc);   // (the host product below runs on the normalised rows)
        }
    }
    StreamArgs a = {}; a.gamma = dG; a.w[0] = dW; a.ws[0] = dS; a.y[0] = dY; a.x = dX; a.groups = 1; a.M = M; a.K = K; a.n = N; a.ldx = K; a.ldy = M; a.tiled = tiled ? 1u : 0u;
    const uint32_t S = argc > 6 ? (uint32_t)atoi(argv[6]) : 1u;
    float* dP = nullptr;
    g_yfinal = dY;
    if (S > 1) { CK(hipMalloc(&dP, (size_t)S * N * M * 4)); CK(hipMemset(dP, 0xFF, (size_t)S * N * M * 4)); a.y[0] = dP; a.ksplit = S; a.ysplit = (uint64_t)N * M; }
    if (getenv("STREAM_WGPCU")) g_wgpcu = atoi(getenv("STREAM_WGPCU")) == 2 ? 2 : 1;
    const uint32_t T = M / 16, ngrp = (uint32_t)(nCU * g_wgpcu) / S, maxt = (T + ngrp - 1) / ngrp;
    printf("M %u K %u N %u: tiles %u, per workgroup <= %u\n", M, K, N, T, maxt);
#define GO(MT) { if (g_dma && N > 112) run<(MT <= 6 ? MT : 6), 8>(a, nCU); else if (g_dma && N > 96) run<(MT <= 6 ? MT : 6), 7>(a, nCU); else if (N <= 16) run<MT, 1>(a, nCU); else if (N <= 32) run<MT, 2>(a, nCU); else if (N <= 48 && g_v2) run<MT, 3>(a, nCU); else if (g_v2 && g_kc == 64 && N > 80 && MT <= 6) run<(MT <= 6 ? MT : 6), 6>(a, nCU); else if (g_v2 && g_kc == 64 && N > 64 && MT <= 6) run<(MT <= 6 ? MT : 6), 5>(a, nCU); else if (g_v2 && g_kc == 64) run<MT, 4>(a, nCU); else run<(MT <= 3 ? MT : 3), 4>(a, nCU); }
    if (maxt <= 1) GO(1) else if (maxt <= 2) GO(2) else if (maxt <= 3) GO(3) else if (maxt <= 4) GO(4) else if (maxt <= 6) GO(6) else GO(8)
    if (getenv("STREAM_CHECK_SKIP")) return 0;   // timing-only runs (the -DSTREAM_PROBE builds compute wrong sums on purpose)
    CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0; const uint32_t CT = (N + 15) / 16;
    std::vector<double> blk((size_t)T * CT, 0.0);
    for (uint32_t c = 0; c < N; ++c) for (uint32_t r = 0; r < M; ++r) {
        double ref = 0; for (uint32_t k = 0; k < K; ++k) ref += (double)W[(size_t)r * K + k] * X[(size_t)c * K + k];
        double e = fabs(ref - Y[(size_t)c * M + r]); if (!(e == e)) e = 1e30;   // a NaN (unwritten output) is an error, not a pass
        worst = e > worst ? e : worst;
        double& b = blk[(size_t)(r / 16) * CT + c / 16]; b = e > b ? e : b;
    }
    printf("max abs err %.3e (values ~ %.1f)\nblock map (rows = 16-row tiles, columns = 16-column tiles; x = wrong):\n", worst, sqrt((double)K) / 3);
    for (uint32_t t = 0; t < T && t < 40; ++t) { printf("  tile %3u ", t); for (uint32_t c = 0; c < CT; ++c) printf("%c", blk[(size_t)t * CT + c] > 1e-3 * sqrt((double)K) ? 'x' : '.'); printf("\n"); }
    // element map of the first wrong block
    for (uint32_t t = 0; t < T; ++t) for (uint32_t c = 0; c < CT; ++c) if (blk[(size_t)t * CT + c] > 1e-3 * sqrt((double)K)) {
        printf("first wrong block: tile %u column tile %u (rows down, columns across; x = wrong)\n", t, c);
        for (uint32_t r = 0; r < 16; ++r) { printf("   "); for (uint32_t j = 0; j < 16 && c * 16 + j < N; ++j) {
            double ref = 0; for (uint32_t k = 0; k < K; ++k) ref += (double)W[(size_t)(t * 16 + r) * K + k] * X[(size_t)(c * 16 + j) * K + k];
            printf("%c", fabs(ref - Y[(size_t)(c * 16 + j) * M + t * 16 + r]) > 1e-3 * sqrt((double)K) ? 'x' : '.'); } printf("\n"); }
        return 0; }
    return 0;
}
