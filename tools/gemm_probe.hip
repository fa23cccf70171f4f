// tools/gemm_probe.hip — k_gemm_mfma alone on the 13B prefill shapes, HIP-event timed; GEMM_ABL=n (compile time) removes parts
// of the kernel to locate where MFMA issue slots are lost.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   -Illama.go_amd/csrc -Iinclude [-DGEMM_ABL=n] -o gemm_probe tools/gemm_probe.hip
#include "kernels_gemm.h"
#include <stdio.h>
#include <vector>
#include <math.h>
using namespace lh;

template <int WN, int WM, int TN, int TM>
static void run_glds(const char* name, uint32_t N, uint32_t M, uint32_t K, uint32_t groups, float* x, float* w, float* y) {
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    size_t lds = (size_t)GST * (BN + BM) * 32 * sizeof(float);
#ifdef PROBE_TWO_PER_CU
    const uint32_t slots = 512;   // two persistent workgroups per CU (needs GST = 2)
#else
    const uint32_t slots = 256;
    if (lds < 82 * 1024) lds = 82 * 1024;
#endif
    auto kern = k_gemm_glds<WN, WM, TN, TM>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GemmArgs a = {};
    a.x = x;
    for (uint32_t g = 0; g < groups; ++g) { a.w[g] = w + (size_t)g * M * K; a.y[g] = y + (size_t)g * N * M; a.r[g] = nullptr; }
    a.groups = groups; a.N = N; a.M = M; a.K = K; a.ldx = K; a.ldy = M; a.ldw = 0;
    const uint32_t tiles = ((N + BN - 1) / BN) * ((M + BM - 1) / BM) * groups;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t grid = tiles < slots ? tiles : slots;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
#ifdef GEMM_CLOCK
    {   // one more launch with the clock stamps of the middle workgroup (-DGEMM_CLOCK): what the shader clock is while this GEMM runs
        unsigned long long* clk; hipMalloc(&clk, 16); hipMemset(clk, 0, 16);
        a.clk = clk;
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, a);
        hipDeviceSynchronize();
        unsigned long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        printf("    shader clock while it runs: %.3f GHz (%llu clocks in %.1f us) -> fp32 matrix peak at this clock %.1f TFLOP/s\n", (double)h[0] / ((double)h[1] * 10.0), h[0],
               (double)h[1] * 0.01, 157.3 * (double)h[0] / ((double)h[1] * 10.0) / 2.4);
        a.clk = nullptr; hipFree(clk);
    }
#endif
    const double fl = 2.0 * N * M * K * groups;
    printf("GLDS(abl %d) %-10s <%d,%d,%d,%d> tile %3dx%3d tiles %5u (%.2f/CU)  %8.1f us  %6.1f TFLOP/s  %.1f %%\n", GEMM_ABL, name, WN, WM, TN, TM, BN, BM, tiles, tiles / 256.0, ms * 1e3,
           fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}

// correctness of k_gemm_glds against k_gemm_mfma on random data (ragged N and M)
static int check(uint32_t N, uint32_t M, uint32_t K) {
    std::vector<float> hx((size_t)N * K), hw((size_t)M * K), y0((size_t)N * M), y1((size_t)N * M);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 23); };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd();
    float *x, *w, *ya, *yb;
    hipMalloc(&x, hx.size() * 4); hipMalloc(&w, hw.size() * 4); hipMalloc(&ya, y0.size() * 4); hipMalloc(&yb, y0.size() * 4);
    hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(ya, 0xff, y0.size() * 4); hipMemset(yb, 0xff, y0.size() * 4);
    GemmArgs a = {};
    a.x = x; a.w[0] = w; a.groups = 1; a.N = N; a.M = M; a.K = K; a.ldx = K; a.ldy = M;
    {
        auto k0 = k_gemm_mfma<2, 2, 2, 2>;
        const size_t lds = 2 * GBK * (129 + 129) * 4;
        hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        a.y[0] = ya;
        hipLaunchKernelGGL(k0, dim3(((N + 127) / 128) * ((M + 127) / 128)), dim3(256), lds, 0, a);
    }
    int bad = 0;
    auto cmp = [&](const char* nm) {
        hipDeviceSynchronize();
        hipMemcpy(y0.data(), ya, y0.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(y1.data(), yb, y0.size() * 4, hipMemcpyDeviceToHost);
        double md = 0, mx = 0;
        for (size_t i = 0; i < y0.size(); ++i) { md = fmax(md, fabs((double)y0[i] - y1[i])); mx = fmax(mx, fabs((double)y0[i])); }
        printf("check %-12s N=%u M=%u K=%u  max|diff| = %.3g (max|y| = %.3g)  %s\n", nm, N, M, K, md, mx, md <= 1e-5 * mx ? "ok" : "MISMATCH");
        if (!(md <= 1e-5 * mx)) bad = 1;
        hipMemset(yb, 0xff, y0.size() * 4);
    };
    a.y[0] = yb;
    {
        auto k1 = k_gemm_glds<4, 1, 1, 5>;
        const size_t lds = GST * (128 + 160) * 32 * 4;
        hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k1, dim3(3), dim3(256), lds, 0, a);
        cmp("glds 128x160");
    }
    {
        auto k1 = k_gemm_glds<2, 2, 2, 2>;
        const size_t lds = GST * (128 + 128) * 32 * 4;
        hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k1, dim3(8), dim3(256), lds, 0, a);
        cmp("glds 128x128");
    }
    {
        auto k1 = k_gemm_glds<2, 2, 2, 1>;
        const size_t lds = GST * (128 + 64) * 32 * 4;
        hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k1, dim3(5), dim3(256), lds, 0, a);
        cmp("glds 128x64");
    }
    hipFree(x); hipFree(w); hipFree(ya); hipFree(yb);
    return bad;
}

template <int WN, int WM, int TN, int TM>
static void run(const char* name, uint32_t N, uint32_t M, uint32_t K, uint32_t groups, float* x, float* w, float* y) {
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    const size_t lds = 2 * GBK * ((BN + 1) + (BM + 1)) * sizeof(float);
    auto kern = k_gemm_mfma<WN, WM, TN, TM>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    GemmArgs a = {};
    a.x = x;
    for (uint32_t g = 0; g < groups; ++g) { a.w[g] = w + (size_t)g * M * K; a.y[g] = y + (size_t)g * N * M; a.r[g] = nullptr; }
    a.groups = groups; a.N = N; a.M = M; a.K = K; a.ldx = K; a.ldy = M; a.ldw = 0;
    const uint32_t tiles = ((N + BN - 1) / BN) * ((M + BM - 1) / BM) * groups;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, 0, a);
    hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), lds, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double fl = 2.0 * N * M * K * groups;
    printf("ABL=%d %-10s <%d,%d,%d,%d> tile %3dx%3d tiles %5u (%.2f/CU)  %8.1f us  %6.1f TFLOP/s  %.1f %%\n", GEMM_ABL, name, WN, WM, TN, TM, BN, BM, tiles, tiles / 256.0,
           ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 / 157.3 * 100);
}

// pseudo-random operands in [-1, 1): the clock the chip holds under matrix load depends on what the multipliers toggle (all-zero operands run cooler)
__global__ void k_fill_rand(float* p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        p[i] = (float)((int)(h >> 8) - (1 << 23)) / (float)(1 << 23);
    }
}

int main() {
#if GEMM_ABL == 0
    if (check(300, 500, 256) | check(128, 160, 64) | check(33, 1000, 96)) return 1;
#endif
    const uint32_t N = 1024, d = 5120, F = 13824;
    float *x, *w, *y;
    hipMalloc(&x, (size_t)N * F * 4);
    hipMalloc(&w, (size_t)3 * F * d * 4);
    hipMalloc(&y, (size_t)3 * N * F * 4);
#ifdef GEMM_CLOCK
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, x, (size_t)N * F, 1u);
    hipLaunchKernelGGL(k_fill_rand, dim3(4096), dim3(256), 0, 0, w, (size_t)3 * F * d, 2u);
    hipDeviceSynchronize();
#else
    hipMemset(x, 0, (size_t)N * F * 4);
    hipMemset(w, 0, (size_t)3 * F * d * 4);
#endif
    run<4, 1, 1, 5>("qkv", N, d, d, 3, x, w, y);
    run<4, 1, 1, 5>("wo", N, d, d, 1, x, w, y);
    run<2, 2, 2, 2>("w1w3", N, F, d, 2, x, w, y);
    run<4, 1, 1, 5>("w2", N, d, F, 1, x, w, y);
    run<2, 2, 2, 2>("wo_128", N, d, d, 1, x, w, y);
#if GEMM_ABL == 0 || GEMM_ABL >= 5
    run_glds<4, 1, 1, 5>("qkv", N, d, d, 3, x, w, y);
    run_glds<4, 1, 1, 5>("qk(2grp)", N, d, d, 2, x, w, y);
    run_glds<4, 1, 1, 5>("M=15360", N, 3 * d, d, 1, x, w, y);
    run_glds<4, 1, 1, 5>("M=10240", N, 2 * d, d, 1, x, w, y);
    run_glds<2, 2, 2, 2>("qkv_128", N, d, d, 3, x, w, y);
    run_glds<4, 1, 1, 5>("wo", N, d, d, 1, x, w, y);
    run_glds<2, 2, 2, 2>("w1w3", N, F, d, 2, x, w, y);
    run_glds<4, 1, 1, 5>("w1w3_160", N, F, d, 2, x, w, y);
    run_glds<4, 1, 1, 5>("w2", N, d, F, 1, x, w, y);
#endif
    return 0;
}
