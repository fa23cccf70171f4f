// tools/kernel_ablate.hip — ablation of the PRODUCT GEMV kernel (csrc/kernels_llama.h) on 7B shapes: bare stream vs
// + row mapping vs + RMSNorm prologue vs + fused epilogue.  Not product code.  Cycles through a 6 GiB weight pool.
#include "../llama.go_amd/csrc/kernels_llama.h"
#include "../llama.go_amd/csrc/kernels_q8.h"
#include "legacy_kernels.h"   // k_gemv, k_gemv_q8: the round-1 kernels the ablations compare against
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float* pool; static const size_t POOL = (size_t)6 << 30;
static hipStream_t st; static hipEvent_t e0, e1; static int nCU;

template <typename K> static void run(const char* label, K kern, GemvArgs a, size_t bytes, int nmat_stride_rows) {
    const size_t lds = 96 * 1024;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    size_t nmat = POOL / bytes; int iters = (int)(3e9 / bytes) + 8;
    auto launch = [&](int i) {
        GemvArgs b = a; float* base = pool + (size_t)(i % nmat) * (bytes / 4);
        const size_t per = bytes / 4 / (a.w[2] ? 3 : a.w[1] ? 2 : 1);
        b.w[0] = base; if (a.w[1]) b.w[1] = base + per; if (a.w[2]) b.w[2] = base + 2 * per;
        hipLaunchKernelGGL(kern, dim3(nCU), dim3(1024), lds, st, b);
    };
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch(i);
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1e3 / iters;
    printf("  %-58s %8.2f us  %7.1f GB/s\n", label, us, bytes / us / 1e3);
    CK(hipGetLastError());
}

int main() {
    CK(hipSetDevice(0)); hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); nCU = p.multiProcessorCount;
    CK(hipMalloc(&pool, POOL));
    { size_t pat = (size_t)16 << 20; std::vector<float> h(pat); unsigned s = 12345;
      for (size_t i = 0; i < pat; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 0.02f; }
      for (size_t off = 0; off < POOL; off += pat * 4) CK(hipMemcpy((char*)pool + off, h.data(), pat * 4, hipMemcpyHostToDevice)); }
    float *x, *g, *y, *q, *kc, *vc, *res; StepParams* sp; double2* rope;
    CK(hipMalloc(&x, 65536 * 4)); CK(hipMalloc(&g, 65536 * 4)); CK(hipMalloc(&y, 65536 * 4)); CK(hipMalloc(&q, 65536 * 4)); CK(hipMalloc(&res, 65536 * 4));
    CK(hipMalloc(&kc, 128 * 4096 * 4)); CK(hipMalloc(&vc, 128 * 4096 * 4)); CK(hipMalloc(&sp, sizeof(StepParams))); CK(hipMalloc(&rope, 256 * 64 * sizeof(double2)));
    std::vector<float> hx(65536); { unsigned s = 7; for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); } }
    CK(hipMemcpy(x, hx.data(), 65536 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hx.data(), 65536 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(res, hx.data(), 65536 * 4, hipMemcpyHostToDevice));
    StepParams hsp = {1, 9, 0, 0}; CK(hipMemcpy(sp, &hsp, sizeof hsp, hipMemcpyHostToDevice));
    std::vector<double2> hr(256 * 64); for (auto& v : hr) { v.x = 0.8; v.y = 0.6; } CK(hipMemcpy(rope, hr.data(), hr.size() * sizeof(double2), hipMemcpyHostToDevice));
    CK(hipStreamCreate(&st)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t d = 4096, F = 11008, V = 32000;
    auto base = [&](uint32_t M, uint32_t K) { GemvArgs a = {}; a.w[0] = pool; a.M = M; a.K = K; a.x = x; a.gamma = g; a.y = y; a.resid = res; a.q_out = q; a.k_cache = kc; a.v_cache = vc; a.rope = rope; a.hd = 128; a.d = d; a.sp = sp; a.rows_per_mat = d; return a; };

    if (getenv("ABL_Q8")) {
        printf("[q8 22016x4096, 36/32 B per weight]\n");
        GemvArgs a = base(2 * F, d); size_t B = (size_t)2 * F * d / 32 * 36;
        a.ws[0] = pool + (size_t)2 * F * d / 4;   // scales somewhere inside the pool (values irrelevant for timing)
        auto runq = [&](const char* label, auto kern) {
            const size_t lds = 96 * 1024;
            CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            size_t nmat = POOL / (B + (1 << 20)); int iters = 60;
            auto launch = [&](int i) { GemvArgs b = a; float* bs = pool + (size_t)(i % nmat) * ((B + (1 << 20)) / 4); b.w[0] = bs; b.ws[0] = bs + (size_t)2 * F * d / 4; b.w[1] = (const float*)((const char*)bs + (size_t)F * d); b.ws[1] = b.ws[0] + (size_t)F * d / 32;
                                       hipLaunchKernelGGL(kern, dim3(nCU), dim3(1024), lds, st, b); };
            for (int i = 0; i < 3; ++i) launch(i);
            CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) launch(i);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / iters;
            printf("  %-58s %8.2f us  %7.1f GB/s\n", label, us, B / us / 1e3); CK(hipGetLastError());
        };
        runq("q8 TPR256 U2 plain/store", k_gemv_q8<1, 2, 256, PRO_PLAIN, EPI_STORE, MAP_SINGLE>);
        runq("q8 TPR256 U4 plain/store", k_gemv_q8<1, 4, 256, PRO_PLAIN, EPI_STORE, MAP_SINGLE>);
        runq("q8 TPR256 U6 plain/store", k_gemv_q8<1, 6, 256, PRO_PLAIN, EPI_STORE, MAP_SINGLE>);
        runq("q8 TPR256 U8 plain/store", k_gemv_q8<1, 8, 256, PRO_PLAIN, EPI_STORE, MAP_SINGLE>);
        runq("q8 TPR1024 KI1 U8 plain/store (256 of 1024 lanes active)", k_gemv_q8<1, 8, 1024, PRO_PLAIN, EPI_STORE, MAP_SINGLE>);
        runq("q8 TPR256 U1 plain/store", k_gemv_q8<1, 1, 256, PRO_PLAIN, EPI_STORE, MAP_SINGLE>);
        runq("q8 TPR256 U2 rmsnorm/silu/pair", k_gemv_q8<1, 2, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>);
        runq("q8 TPR256 U4 rmsnorm/silu/pair", k_gemv_q8<1, 4, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>);
        printf("done\n");
        return 0;
    }
    if (getenv("ABL_Q8S")) {   // round 2: the scalar-addressed two-set kernel (k_gemv_q8s): rows in flight, waves per row, workgroups per CU
        struct Shape { const char* name; uint32_t M, K; int pair; } shapes[] = {{"w1w3 22016x4096", 2 * F, d, 1}, {"wo 4096x4096", d, d, 0}, {"w2 4096x11008", d, F, 0}};
        for (const Shape& sh : shapes) {
            printf("[q8s %s, 36/32 B per weight]\n", sh.name);
            GemvArgs a = base(sh.M, sh.K);
            const size_t QB = (size_t)sh.M * sh.K, B = QB / 32 * 36;
            auto runq = [&](const char* label, auto kern, int wgpcu, int threads) {
                const size_t lds = wgpcu == 2 ? 64 * 1024 : 96 * 1024;
                CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
                const size_t slot = (B + (1 << 20)) & ~(size_t)4095;
                size_t nmat = POOL / slot; int iters = 80;
                auto launch = [&](int i) {
                    GemvArgs b = a; const char* bs = (const char*)pool + (size_t)(i % nmat) * slot;
                    b.w[0] = (const float*)bs; b.ws[0] = (const float*)(bs + QB);
                    if (sh.pair) { b.w[1] = (const float*)(bs + QB / 2); b.ws[1] = (const float*)(bs + QB + QB / 32 * 4 / 2); }
                    hipLaunchKernelGGL(kern, dim3(nCU * wgpcu), dim3(threads), lds, st, b); };
                for (int i = 0; i < 3; ++i) launch(i);
                CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) launch(i);
                CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double us = ms * 1e3 / iters;
                printf("  %-58s %8.2f us  %7.1f GB/s\n", label, us, B / us / 1e3); CK(hipGetLastError());
            };
            if (sh.K == d && sh.pair) {
                runq("old k_gemv_q8 TPR256 U2 rmsnorm/silu/pair", k_gemv_q8<1, 2, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR256 U1 rmsnorm/silu/pair", k_gemv_q8s<1, 1, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR256 U2 rmsnorm/silu/pair", k_gemv_q8s<1, 2, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR256 U3 rmsnorm/silu/pair", k_gemv_q8s<1, 3, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR256 U4 rmsnorm/silu/pair", k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR256 U6 rmsnorm/silu/pair", k_gemv_q8s<1, 6, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR128 KI2 U2 rmsnorm/silu/pair", k_gemv_q8s<2, 2, 128, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR128 KI2 U4 rmsnorm/silu/pair", k_gemv_q8s<2, 4, 128, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR64 KI4 U2 rmsnorm/silu/pair", k_gemv_q8s<4, 2, 64, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR64 KI4 U4 rmsnorm/silu/pair", k_gemv_q8s<4, 4, 64, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 1, 1024);
                runq("q8s TPR256 U2 plain/store/pair (bare stream)", k_gemv_q8s<1, 2, 256, PRO_PLAIN, EPI_STORE, MAP_PAIR>, 1, 1024);
                runq("q8s TPR256 U4 plain/store/pair (bare stream)", k_gemv_q8s<1, 4, 256, PRO_PLAIN, EPI_STORE, MAP_PAIR>, 1, 1024);
                runq("q8s TPR256 U2 rmsnorm/silu/pair, 2 wg/CU", k_gemv_q8s<1, 2, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 2, 1024);
                runq("q8s TPR256 U4 rmsnorm/silu/pair, 2 wg/CU", k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, 2, 1024);
            } else if (sh.K == d) {
                runq("old k_gemv_q8 TPR256 U2 plain/resid", k_gemv_q8<1, 2, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
                runq("q8s TPR256 U1 plain/resid", k_gemv_q8s<1, 1, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
                runq("q8s TPR256 U2 plain/resid", k_gemv_q8s<1, 2, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
                runq("q8s TPR256 U4 plain/resid", k_gemv_q8s<1, 4, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
                runq("q8s TPR64 KI4 U1 plain/resid", k_gemv_q8s<4, 1, 64, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
            } else {
                runq("old k_gemv_q8 TPR1024 KI1 U2 plain/resid", k_gemv_q8<1, 2, 1024, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
                runq("q8s TPR1024 KI1 U2 plain/resid", k_gemv_q8s<1, 2, 1024, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
                runq("q8s TPR1024 KI1 U4 plain/resid", k_gemv_q8s<1, 4, 1024, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
                runq("q8s TPR256 KI3 U2 plain/resid", k_gemv_q8s<3, 2, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
                runq("q8s TPR256 KI3 U4 plain/resid", k_gemv_q8s<3, 4, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
                runq("q8s TPR512 KI2 U2 plain/resid", k_gemv_q8s<2, 2, 512, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, 1, 1024);
            }
        }
        printf("done\n");
        return 0;
    }
    printf("[w1w3 2x11008x4096]\n");
    { GemvArgs a = base(2 * F, d); size_t B = (size_t)2 * F * d * 4;
      run("U4 plain/store/single (bare stream)", k_gemv<1, 4, 1024, PRO_PLAIN, EPI_STORE, MAP_SINGLE>, a, B, 0);
      GemvArgs b = a; b.w[1] = pool + 1;
      run("U4 plain/store/pair", k_gemv<1, 4, 1024, PRO_PLAIN, EPI_STORE, MAP_PAIR>, b, B, 0);
      run("U4 rmsnorm/store/pair", k_gemv<1, 4, 1024, PRO_RMSNORM, EPI_STORE, MAP_PAIR>, b, B, 0);
      run("U4 plain/silu/pair", k_gemv<1, 4, 1024, PRO_PLAIN, EPI_SILU_MUL, MAP_PAIR>, b, B, 0);
      run("U4 rmsnorm/silu/pair (product)", k_gemv<1, 4, 1024, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, b, B, 0);
      run("U2 rmsnorm/silu/pair", k_gemv<1, 2, 1024, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, b, B, 0);
      run("U6 rmsnorm/silu/pair", k_gemv<1, 6, 1024, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, b, B, 0);
      run("U8 rmsnorm/silu/pair", k_gemv<1, 8, 1024, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, b, B, 0); }
    printf("[qkv 3x4096x4096]\n");
    { GemvArgs a = base(3 * d, d); size_t B = (size_t)3 * d * d * 4; GemvArgs b = a; b.w[1] = pool + 1; b.w[2] = pool + 2;
      run("U4 plain/store/single (bare stream)", k_gemv<1, 4, 1024, PRO_PLAIN, EPI_STORE, MAP_SINGLE>, a, B, 0);
      run("U4 plain/store/block", k_gemv<1, 4, 1024, PRO_PLAIN, EPI_STORE, MAP_BLOCK>, b, B, 0);
      run("U4 rmsnorm/store/block", k_gemv<1, 4, 1024, PRO_RMSNORM, EPI_STORE, MAP_BLOCK>, b, B, 0);
      run("U4 plain/rope/block", k_gemv<1, 4, 1024, PRO_PLAIN, EPI_QKV_ROPE, MAP_BLOCK>, b, B, 0);
      run("U4 rmsnorm/rope/block (product)", k_gemv<1, 4, 1024, PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK>, b, B, 0); }
    printf("[wo 4096x4096]\n");
    { GemvArgs a = base(d, d); size_t B = (size_t)d * d * 4;
      run("U4 plain/store/single (bare stream)", k_gemv<1, 4, 1024, PRO_PLAIN, EPI_STORE, MAP_SINGLE>, a, B, 0);
      run("U4 plain/resid/single (product)", k_gemv<1, 4, 1024, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, a, B, 0);
      run("U2 plain/resid/single", k_gemv<1, 2, 1024, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, a, B, 0); }
    printf("[w2 4096x11008]\n");
    { GemvArgs a = base(d, F); size_t B = (size_t)d * F * 4;
      run("KI3 U2 plain/store/single (bare stream)", k_gemv<3, 2, 1024, PRO_PLAIN, EPI_STORE, MAP_SINGLE>, a, B, 0);
      run("KI3 U2 plain/resid/single (product)", k_gemv<3, 2, 1024, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, a, B, 0);
      run("KI3 U4 plain/resid/single", k_gemv<3, 4, 1024, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, a, B, 0); }
    printf("[lmhead 32000x4096]\n");
    { GemvArgs a = base(V, d); size_t B = (size_t)V * d * 4;
      run("U4 plain/store/single (bare stream)", k_gemv<1, 4, 1024, PRO_PLAIN, EPI_STORE, MAP_SINGLE>, a, B, 0);
      run("U4 rmsnorm/store/single (product)", k_gemv<1, 4, 1024, PRO_RMSNORM, EPI_STORE, MAP_SINGLE>, a, B, 0); }
    printf("done\n");
    return 0;
}
