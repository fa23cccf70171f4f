// tools/q8s_phase_probe.hip — what the phases of the block-int8 decode GEMV (k_gemv_q8s, csrc/kernels_q8.h) cost on the 7B launches as shipped
// (256-thread workgroups): the bare stream against + RMSNorm prologue, + fused epilogue, rows in flight, three matrices against the same bytes as one
// (MAP_BLOCK's row addressing) - and the fp32 twin of the last comparison.  Timing only (weights rotate through a pool larger than the Infinity Cache;
// values are irrelevant).  Not product code.  The variants measured with it in round 6 and not kept (both register sets requested in front of the norm,
// epilogue operands requested behind the first rows) are recorded in profiles/r06_q8s_phase_probe.txt.
#include "../llama.go_amd/csrc/kernels_llama.h"
#include "../llama.go_amd/csrc/kernels_q8.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace lh;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static char* pool; static const size_t POOL = (size_t)3 << 30;
static hipStream_t st; static hipEvent_t e0, e1; static int nCU;
static bool g_hostpart = false;
static size_t g_skew = 0;
static bool g_separate = false;   // every matrix slot its own hipMalloc (as the model's weights are) instead of a slice of one 3 GB allocation: page-table fragments / TLB reach   // bytes the planes of matrix m are shifted by (m * skew): does it matter WHERE the second matrix of a pair lies?   // the row blocks of the workgroups from the host's quotient / remainder (wg_row_block) instead of the kernels' own divisions

int main() {
    CK(hipSetDevice(0)); hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); nCU = p.multiProcessorCount;
    CK(hipMalloc(&pool, POOL));
    { size_t pat = (size_t)16 << 20; std::vector<float> h(pat / 4); unsigned s = 12345;
      for (size_t i = 0; i < h.size(); ++i) { s = s * 1664525u + 1013904223u; h[i] = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 0.02f; }
      for (size_t off = 0; off < POOL; off += pat) CK(hipMemcpy(pool + off, h.data(), pat, hipMemcpyHostToDevice)); }
    float *x, *g, *y, *q, *kc, *vc, *res; StepParams* sp; double2* rope;
    CK(hipMalloc(&x, 65536 * 4)); CK(hipMalloc(&g, 65536 * 4)); CK(hipMalloc(&y, 65536 * 4)); CK(hipMalloc(&q, 65536 * 4)); CK(hipMalloc(&res, 65536 * 4));
    CK(hipMalloc(&kc, 128 * 4096 * 4)); CK(hipMalloc(&vc, 128 * 4096 * 4)); CK(hipMalloc(&sp, sizeof(StepParams))); CK(hipMalloc(&rope, 256 * 64 * sizeof(double2)));
    std::vector<float> hx(65536); { unsigned s = 7; for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); } }
    CK(hipMemcpy(x, hx.data(), 65536 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(g, hx.data(), 65536 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(res, hx.data(), 65536 * 4, hipMemcpyHostToDevice));
    StepParams hsp = {1, 9, 0, 0}; CK(hipMemcpy(sp, &hsp, sizeof hsp, hipMemcpyHostToDevice));
    std::vector<double2> hr(256 * 64); for (auto& v : hr) { v.x = 0.8; v.y = 0.6; } CK(hipMemcpy(rope, hr.data(), hr.size() * sizeof(double2), hipMemcpyHostToDevice));
    CK(hipStreamCreate(&st)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const uint32_t d = 4096, F = 11008, V = 32000;
    auto base = [&](uint32_t M, uint32_t K) { GemvArgs a = {}; a.M = M; a.K = K; a.x = x; a.gamma = g; a.y = y; a.resid = res; a.q_out = q; a.k_cache = kc; a.v_cache = vc; a.rope = rope; a.hd = 128; a.d = d; a.sp = sp; a.rows_per_mat = d; return a; };
    // mats matrices of rows x K quants each, then their scale planes
    auto runq = [&](const char* label, auto kern, GemvArgs a, uint32_t mats, uint32_t rows, uint32_t K) {
        const size_t QB = (size_t)rows * K, SB = QB / 32 * 4, B = (size_t)mats * (QB + SB);
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        const size_t slot = (B + (8 << 20)) & ~(size_t)4095;
        const size_t nmat = POOL / slot; const int iters = 200;
        std::vector<char*> sep;
        if (g_separate) { sep.resize(nmat); for (size_t k = 0; k < nmat; ++k) CK(hipMalloc((void**)&sep[k], slot)); }
        auto launch = [&](int i) {
            GemvArgs b = a; const char* bs = g_separate ? sep[(size_t)i % nmat] : pool + (size_t)(i % nmat) * slot;
            for (uint32_t m = 0; m < mats; ++m) { b.w[m] = (const float*)(bs + m * (QB + g_skew)); b.ws[m] = (const float*)(bs + mats * (QB + g_skew) + m * (SB + g_skew)); }
            if (g_hostpart) { b.wg_q = (b.M / 2) / (uint32_t)nCU; b.wg_r = (b.M / 2) % (uint32_t)nCU; }
            hipLaunchKernelGGL(kern, dim3(nCU), dim3(256), 96 * 1024, st, b); };
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            for (int i = 0; i < 5; ++i) launch(i);
            CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) launch(i);
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); const double us = ms * 1e3 / iters;
            best = us < best ? us : best;
        }
        printf("  %-66s %8.2f us  %7.1f GB/s\n", label, best, B / best / 1e3); CK(hipGetLastError());
        for (char* q : sep) CK(hipFree(q));
    };
    if (getenv("PROBE_SEPARATE")) {
        // the same launches out of ONE 3 GB allocation and out of one hipMalloc per matrix slot (the pool is freed first: the separate slots need the room)
        g_hostpart = true;
        GemvArgs a1 = base(2 * F, d), aq = base(3 * d, d);
        for (int sepa = 0; sepa < 2; ++sepa) {
            g_separate = sepa == 1;
            if (g_separate) { CK(hipFree(pool)); pool = nullptr; }
            printf("===== weights: %s\n", g_separate ? "one hipMalloc per matrix slot (as the model allocates)" : "slices of one 3 GB hipMalloc");
            runq("w1|w3 rmsnorm / silu*mul", k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR, 256>, a1, 2, F, d);
            runq("wq|wk|wv rmsnorm / rope + cache", k_gemv_q8s<1, 6, 256, PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK, 256>, aq, 3, d, d);
            runq("wo plain / + residual", k_gemv_q8s<1, 4, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, d), 1, d, d);
            runq("w2 plain / + residual", k_gemv_q8s<3, 2, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, F), 1, d, F);
            runq("lm_head rmsnorm / store", k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_STORE, MAP_SINGLE, 256>, base(V, d), 1, V, d);
        }
        printf("done\n");
        return 0;
    }
    for (int pass = 0; pass < 2; ++pass) {
    g_hostpart = pass == 1;
    printf("===== row blocks: %s\n", g_hostpart ? "host quotient / remainder (wg_row_block, as shipped)" : "two 64-bit divisions in the kernel (until round 6)");
    printf("[w1|w3 2 x 11008 x 4096 block-int8: k_gemv_q8s<1, 4, 256, ., ., MAP_PAIR, 256>]\n");
    { GemvArgs a = base(2 * F, d);
      runq("plain / store (bare stream)", k_gemv_q8s<1, 4, 256, PRO_PLAIN, EPI_STORE, MAP_PAIR, 256>, a, 2, F, d);
      runq("rmsnorm / store", k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_STORE, MAP_PAIR, 256>, a, 2, F, d);
      runq("plain / silu*mul", k_gemv_q8s<1, 4, 256, PRO_PLAIN, EPI_SILU_MUL, MAP_PAIR, 256>, a, 2, F, d);
      runq("rmsnorm / silu*mul (shipped)", k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR, 256>, a, 2, F, d);
      runq("rmsnorm / store as ONE matrix of 22016 rows (MAP_SINGLE)", k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_STORE, MAP_SINGLE, 256>, base(2 * F, d), 1, 2 * F, d);
      runq("plain / store as ONE matrix of 22016 rows (MAP_SINGLE)", k_gemv_q8s<1, 4, 256, PRO_PLAIN, EPI_STORE, MAP_SINGLE, 256>, base(2 * F, d), 1, 2 * F, d);
      for (size_t sk : {(size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)(1 << 20) + 4096, (size_t)(2 << 20)}) {
          g_skew = sk; char lb[96]; snprintf(lb, sizeof lb, "rmsnorm / silu*mul, w3's planes shifted by %zu bytes", sk);
          runq(lb, k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR, 256>, a, 2, F, d);
      }
      g_skew = 0;
      runq("rmsnorm / silu*mul, U = 2", k_gemv_q8s<1, 2, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR, 256>, a, 2, F, d);
      runq("rmsnorm / silu*mul, U = 6", k_gemv_q8s<1, 6, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR, 256>, a, 2, F, d);
      runq("rmsnorm / silu*mul, U = 8", k_gemv_q8s<1, 8, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR, 256>, a, 2, F, d); }
    printf("[wq|wk|wv 3 x 4096 x 4096 block-int8: k_gemv_q8s<1, 6, 256, ., ., MAP_BLOCK, 256>]\n");
    { GemvArgs a = base(3 * d, d);
      runq("plain / store (bare stream)", k_gemv_q8s<1, 6, 256, PRO_PLAIN, EPI_STORE, MAP_BLOCK, 256>, a, 3, d, d);
      runq("rmsnorm / store", k_gemv_q8s<1, 6, 256, PRO_RMSNORM, EPI_STORE, MAP_BLOCK, 256>, a, 3, d, d);
      runq("plain / rope + cache", k_gemv_q8s<1, 6, 256, PRO_PLAIN, EPI_QKV_ROPE, MAP_BLOCK, 256>, a, 3, d, d);
      runq("rmsnorm / rope + cache (shipped)", k_gemv_q8s<1, 6, 256, PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK, 256>, a, 3, d, d);
      runq("rmsnorm / rope + cache, U = 4", k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK, 256>, a, 3, d, d);
      runq("rmsnorm / rope + cache, U = 8", k_gemv_q8s<1, 8, 256, PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK, 256>, a, 3, d, d);
      runq("rmsnorm / rope + cache as ONE matrix of 12288 rows (MAP_SINGLE)", k_gemv_q8s<1, 6, 256, PRO_RMSNORM, EPI_STORE, MAP_SINGLE, 256>, base(3 * d, d), 1, 3 * d, d); }
    printf("[wo 4096 x 4096 / w2 4096 x 11008 block-int8, plain / + residual]\n");
    { runq("wo <1, 4, 256> shipped", k_gemv_q8s<1, 4, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, d), 1, d, d);
      runq("wo <1, 8, 256>", k_gemv_q8s<1, 8, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, d), 1, d, d);
      runq("wo <1, 6, 256>", k_gemv_q8s<1, 6, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, d), 1, d, d);
      runq("wo <1, 2, 256>", k_gemv_q8s<1, 2, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, d), 1, d, d);
      runq("w2 <3, 3, 256>", k_gemv_q8s<3, 3, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, F), 1, d, F);
      runq("w2 <3, 4, 256>", k_gemv_q8s<3, 4, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, F), 1, d, F);
      runq("w2 <3, 1, 256>", k_gemv_q8s<3, 1, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, F), 1, d, F);
      runq("w2 <3, 2, 256> shipped", k_gemv_q8s<3, 2, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>, base(d, F), 1, d, F); }
    printf("[lm_head 32000 x 4096 block-int8]\n");
    { runq("rmsnorm / store <1, 4, 256> shipped", k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_STORE, MAP_SINGLE, 256>, base(V, d), 1, V, d);
      runq("plain / store <1, 4, 256>", k_gemv_q8s<1, 4, 256, PRO_PLAIN, EPI_STORE, MAP_SINGLE, 256>, base(V, d), 1, V, d); }
    // the four launches of a layer in ROTATION (as the decode step issues them) against the sum of the four alone: does a kernel pay for finding its code
    // cold (five different kernels alternate in the model, each of them a few KB of instructions) or its operands in another XCD's L2?
    printf("[block-int8 layer rotation wq|wk|wv -> wo -> w1|w3 -> w2, per ROUND of four launches]\n");
    { GemvArgs aq = base(3 * d, d), ao = base(d, d), a1 = base(2 * F, d), a2 = base(d, F);
      const size_t QBq = (size_t)d * d, QB1 = (size_t)F * d;
      const size_t Bq = 3 * (QBq + QBq / 8), Bo = QBq + QBq / 8, B1 = 2 * (QB1 + QB1 / 8), B2 = QB1 + QB1 / 8, Ball = Bq + Bo + B1 + B2;
      const size_t slot = (Ball + (1 << 20)) & ~(size_t)4095, nmat = POOL / slot;
      auto kq = k_gemv_q8s<1, 6, 256, PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK, 256>; auto ko = k_gemv_q8s<1, 4, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>;
      auto k1 = k_gemv_q8s<1, 4, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR, 256>; auto k2 = k_gemv_q8s<3, 2, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE, 256>;
      for (auto kp : {(const void*)kq, (const void*)ko, (const void*)k1, (const void*)k2}) CK(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
      auto part = [&](GemvArgs& b) { if (g_hostpart) { b.wg_q = (b.M / 2) / (uint32_t)nCU; b.wg_r = (b.M / 2) % (uint32_t)nCU; } };
      auto round = [&](int i, int which) {   // which: bit mask of the launches to issue
          const char* bs = pool + (size_t)(i % nmat) * slot;
          if (which & 1) { GemvArgs b = aq; for (int m = 0; m < 3; ++m) { b.w[m] = (const float*)(bs + m * QBq); b.ws[m] = (const float*)(bs + 3 * QBq + m * (QBq / 8)); } part(b); hipLaunchKernelGGL(kq, dim3(nCU), dim3(256), 96 * 1024, st, b); }
          bs += Bq;
          if (which & 2) { GemvArgs b = ao; b.w[0] = (const float*)bs; b.ws[0] = (const float*)(bs + QBq); part(b); hipLaunchKernelGGL(ko, dim3(nCU), dim3(256), 96 * 1024, st, b); }
          bs += Bo;
          if (which & 4) { GemvArgs b = a1; for (int m = 0; m < 2; ++m) { b.w[m] = (const float*)(bs + m * QB1); b.ws[m] = (const float*)(bs + 2 * QB1 + m * (QB1 / 8)); } part(b); hipLaunchKernelGGL(k1, dim3(nCU), dim3(256), 96 * 1024, st, b); }
          bs += B1;
          if (which & 8) { GemvArgs b = a2; b.w[0] = (const float*)bs; b.ws[0] = (const float*)(bs + QB1); part(b); hipLaunchKernelGGL(k2, dim3(nCU), dim3(256), 96 * 1024, st, b); }
      };
      auto timew = [&](const char* label, int which) {
          double best = 1e30;
          for (int rep = 0; rep < 3; ++rep) {
              for (int i = 0; i < 5; ++i) round(i, which);
              CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st));
              for (int i = 0; i < 100; ++i) round(i, which);
              CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
              float ms; CK(hipEventElapsedTime(&ms, e0, e1)); const double us = ms * 1e3 / 100; best = us < best ? us : best;
          }
          printf("  %-66s %8.2f us\n", label, best); return best;
      };
      const double tq = timew("wq|wk|wv alone", 1), to = timew("wo alone", 2), t1 = timew("w1|w3 alone", 4), t2 = timew("w2 alone", 8);
      const double tr = timew("the four in rotation", 15);
      printf("  sum of the four alone %.2f us, in rotation %.2f us: %+.2f us per round\n", tq + to + t1 + t2, tr, tr - (tq + to + t1 + t2)); }
    // fp32 twin: does k_gemv_sa pay for MAP_BLOCK's row addressing too?  (rows are 4x longer: the scalar work per row weighs a quarter)
    printf("[fp32 wq|wk|wv 3 x 4096 x 4096: k_gemv_sa<4, 2, 256, ., ., ., 256>]\n");
    { auto runf = [&](const char* label, auto kern, GemvArgs a, uint32_t mats, uint32_t rows, uint32_t K) {
          const size_t MB = (size_t)rows * K * 4, B = (size_t)mats * MB;
          CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
          const size_t slot = (B + (1 << 20)) & ~(size_t)4095;
          const size_t nmat = POOL / slot; const int iters = 100;
          auto launch = [&](int i) {
              GemvArgs b = a; const char* bs = pool + (size_t)(i % nmat) * slot;
              for (uint32_t m = 0; m < mats; ++m) b.w[m] = (const float*)(bs + m * MB);
              if (g_hostpart) { b.wg_q = (b.M / 2) / (uint32_t)nCU; b.wg_r = (b.M / 2) % (uint32_t)nCU; }
              hipLaunchKernelGGL(kern, dim3(nCU), dim3(256), 96 * 1024, st, b); };
          double best = 1e30;
          for (int rep = 0; rep < 3; ++rep) {
              for (int i = 0; i < 5; ++i) launch(i);
              CK(hipStreamSynchronize(st)); CK(hipEventRecord(e0, st));
              for (int i = 0; i < iters; ++i) launch(i);
              CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
              float ms; CK(hipEventElapsedTime(&ms, e0, e1)); const double us = ms * 1e3 / iters;
              best = us < best ? us : best;
          }
          printf("  %-66s %8.2f us  %7.1f GB/s\n", label, best, B / best / 1e3); CK(hipGetLastError());
      };
      runf("rmsnorm / rope + cache, MAP_BLOCK (shipped)", k_gemv_sa<4, 2, 256, PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK>, base(3 * d, d), 3, d, d);
      runf("rmsnorm / store, MAP_BLOCK", k_gemv_sa<4, 2, 256, PRO_RMSNORM, EPI_STORE, MAP_BLOCK>, base(3 * d, d), 3, d, d);
      runf("rmsnorm / store as ONE matrix of 12288 rows (MAP_SINGLE)", k_gemv_sa<4, 2, 256, PRO_RMSNORM, EPI_STORE, MAP_SINGLE>, base(3 * d, d), 1, 3 * d, d);
      runf("plain / store, MAP_BLOCK", k_gemv_sa<4, 2, 256, PRO_PLAIN, EPI_STORE, MAP_BLOCK>, base(3 * d, d), 3, d, d);
      runf("plain / store, MAP_SINGLE", k_gemv_sa<4, 2, 256, PRO_PLAIN, EPI_STORE, MAP_SINGLE>, base(3 * d, d), 1, 3 * d, d);
      runf("fp32 wo 4096 x 4096 plain / + residual (shipped)", k_gemv_sa<4, 2, 256, PRO_PLAIN, EPI_RESID, MAP_SINGLE>, base(d, d), 1, d, d);
      runf("fp32 w1|w3 rmsnorm / silu*mul (shipped)", k_gemv_sa<4, 2, 256, PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>, base(2 * F, d), 2, F, d); }
    }
    printf("done\n");
    return 0;
}
