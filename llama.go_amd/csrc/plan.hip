// csrc/plan.hip — fused LLaMA plan executor: decode step (hipGraph-replayed), small-N prefill, pipeline
// stages, device-resident greedy loop, per-kernel event profiling.  Entry points: lh_llama_* (llamahip.h).
// Follows the layer schedule of llama.Eval (pkg/llama/llama.go:246-384).
#include "plan.h"
#include "kernels_llama.h"
#include "kernels_gemm.h"
#include "kernels_q8.h"
#include "kernels_skinny.h"
#include "kernels_attn.h"
#include "kernels_sample.h"
#include "kernels_stream.h"
#include "kernels_stream_q8b.h"
#include "kernels_stream_b9.h"
#include "kernels_gemm_b9.h"
#include "kernels_rows.h"
#include <math.h>
#include <string.h>
#include <algorithm>

namespace lh {

static thread_local bool g_prepare_only = false;  // set kernel attributes without launching (before graph capture)
static thread_local const char* g_only = nullptr;   // lh_llama_profile_decode: launch only the kernels of this name (timing pass)
static inline bool skip_launch(const char* name) { return g_prepare_only || (g_only && strcmp(g_only, name) != 0); }

static constexpr int TH = 1024;
static constexpr size_t FAT_LDS = 96 * 1024;  // > 80 KiB: one fat workgroup per CU

struct ProfSink {
    struct Rec { const char* name; uint64_t bytes; hipEvent_t e0, e1; };
    std::vector<Rec> recs;
    bool on = false;
};
static thread_local ProfSink* g_prof = nullptr;

// LLAMAHIP_TRACE=1: print every launch and synchronise after it (debugging aid: the last line names a faulting kernel)
static bool trace_on() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("LLAMAHIP_TRACE"); v = (e && e[0] == '1') ? 1 : 0; }
    return v == 1;
}
struct TraceScope {
    hipStream_t st;
    const char* name;
    TraceScope(hipStream_t s, const char* n) : st(s), name(n) { if (trace_on() && !g_prepare_only) { fprintf(stderr, "[lh] launch %s\n", n); fflush(stderr); } }
    ~TraceScope() {
        if (trace_on() && !g_prepare_only) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            hipStreamIsCapturing(st, &cs);
            if (cs == hipStreamCaptureStatusNone) { hipError_t e = hipStreamSynchronize(st); fprintf(stderr, "[lh]   done %s: %s\n", name, hipGetErrorString(e)); fflush(stderr); }
        }
    }
};

struct ProfScope {
    hipStream_t st;
    ProfSink::Rec rec;
    bool on;
    TraceScope tr;
    ProfScope(hipStream_t s, const char* name, uint64_t bytes) : st(s), on(g_prof && g_prof->on && !g_prepare_only), tr(s, name) {
        if (on) {
            rec.name = name; rec.bytes = bytes;
            hipEventCreate(&rec.e0); hipEventCreate(&rec.e1);
            hipEventRecord(rec.e0, st);
        }
    }
    ~ProfScope() {
        if (on) { hipEventRecord(rec.e1, st); g_prof->recs.push_back(rec); }
    }
};

// Parks the stream for `ticks` of the 100 MHz realtime counter: lh_llama_profile_decode queues its whole event/kernel sequence
// behind it, so host launch latency (event creation, API calls) does not leak into the per-kernel event intervals.
__global__ void k_park(uint64_t ticks) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

template <typename KernT>
static int set_lds_once(lh_ctx* ctx, KernT kern, size_t lds, bool* flags) {
    if (!flags[ctx->device & 15]) {
        LH_HIP(ctx, hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        flags[ctx->device & 15] = true;
    }
    return 0;
}

// "This kernel is not built for the shape - take the next one": positive, so it can never be mistaken for an LH_E* code (all negative).
// Callers fall back ONLY on ST_NA; every other non-zero value is a real failure and is returned as it is.
static constexpr int ST_NA = 1;

template <int KI, int U, int PRO, int EPI, int MAP, int THR = TH>
static int launch_gemv(lh_ctx* ctx, const GemvArgs& a, const char* name, uint64_t bytes) {
    static bool flags[16] = {};
    int rc = set_lds_once(ctx, k_gemv_sa<KI, U, THR, PRO, EPI, MAP>, FAT_LDS, flags);
    if (rc) return rc;
    if (skip_launch(name)) return 0;
    ProfScope ps(ctx->stream, name, bytes);
    GemvArgs b = a;
    b.wg_q = (a.M / 2) / (uint32_t)ctx->ds->num_cu; b.wg_r = (a.M / 2) % (uint32_t)ctx->ds->num_cu;   // wg_row_block
    LH_LAUNCH((k_gemv_sa<KI, U, THR, PRO, EPI, MAP>), dim3(ctx->ds->num_cu), dim3(THR), FAT_LDS, ctx->stream, b);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

template <int KI, int U, int TPR, int PRO, int EPI, int MAP, int THR = TH>
static int launch_gemv_q8(lh_ctx* ctx, const GemvArgs& a, const char* name, uint64_t bytes) {
    static bool flags[16] = {};
    int rc = set_lds_once(ctx, k_gemv_q8s<KI, U, TPR, PRO, EPI, MAP, THR>, FAT_LDS, flags);   // one fat workgroup per CU (LDS request > 80 KiB)
    if (rc) return rc;
    if (skip_launch(name)) return 0;
    ProfScope ps(ctx->stream, name, bytes);
    GemvArgs b = a;
    b.wg_q = (a.M / 2) / (uint32_t)ctx->ds->num_cu; b.wg_r = (a.M / 2) % (uint32_t)ctx->ds->num_cu;   // wg_row_block
    LH_LAUNCH((k_gemv_q8s<KI, U, TPR, PRO, EPI, MAP, THR>), dim3(ctx->ds->num_cu), dim3(THR), FAT_LDS, ctx->stream, b);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// block-int8 GEMV: pick threads-per-row so the 16-byte chunks of a row fill the lanes (K = 4096 -> 256 threads x 4 rows side by side)
template <int PRO, int EPI, int MAP>
static int gemv_q8(lh_ctx* ctx, const GemvArgs& a, const char* name) {
    if (a.K % 32 || ((EPI == EPI_QKV_ROPE || EPI == EPI_SILU_MUL) && a.M % 2))
        LH_FAIL(ctx, LH_ESHAPE, "gemv_q8 %s: K=%u must be a multiple of 32%s", name, a.K, a.M % 2 ? " and the row count even" : "");
    if ((uint64_t)a.M / ctx->ds->num_cu + 4 > (uint64_t)TH) LH_FAIL(ctx, LH_EUNSUPPORTED, "gemv_q8 %s: M=%u exceeds %d rows per workgroup", name, a.M, TH - 4);
    const uint32_t K16 = a.K / 16;
    const uint64_t bytes = (uint64_t)a.M * a.K / 32 * 36;
    // Rows in flight per row group (x2 register sets), tools/kernel_ablate ABL_Q8S on the scalar-addressed kernel (profiles/r02_q8s_ablation.txt):
    // 22016 x 4096: U = 1 / 2 / 3 / 4 / 6 -> 21.6 / 22.2 / 21.9 / 21.3 / 21.7 us back to back, but inside the decode graph U = 4 lost to U = 2
    // (501 vs 513 tok/s same-day); 4096 x 4096: U = 1 / 2 / 4 -> 5.8 / 6.0 / 7.6 us (16 rows per CU:
    // deeper batches only add dummy loads); K = 11008: one row across the whole workgroup, U = 2 (U = 4: 13.0 vs 11.6 us)
    const uint32_t rows_wg = a.M / (uint32_t)ctx->ds->num_cu;
    // Round 3: 256-thread workgroups (ONE row group of four waves, like the fp32 stream) with U rows in flight per register set instead of
    // 1024 threads = four row groups: 489 -> 534-536 tok/s on 7B (profiles/r03_q8_workgroup_256.txt; per kernel w1|w3 24.8 -> 22.1 us, w2
    // 14.9 -> 13.4, wo 9.0 -> 8.1, lm_head 34 -> 29, wq|wk|wv 17.4 -> 16.1 with six rows).  U swept 2..8 per shape: 4 (wq|wk|wv: 6), K = 11008: 2.
    // Same threads per row, same per-lane arithmetic, same cross-wave order: results bit-identical to the 1024-thread launch.
    if (rows_wg + 4 <= 250) {
        if (K16 <= 256) return MAP == MAP_BLOCK ? launch_gemv_q8<1, 6, 256, PRO, EPI, MAP, 256>(ctx, a, name, bytes) : launch_gemv_q8<1, 4, 256, PRO, EPI, MAP, 256>(ctx, a, name, bytes);
        if (K16 > 512 && K16 <= 768) return launch_gemv_q8<3, 2, 256, PRO, EPI, MAP, 256>(ctx, a, name, bytes);
    }
    if (K16 <= 256) return rows_wg >= 32 ? launch_gemv_q8<1, 2, 256, PRO, EPI, MAP>(ctx, a, name, bytes) : launch_gemv_q8<1, 1, 256, PRO, EPI, MAP>(ctx, a, name, bytes);
    if (K16 <= 512) return launch_gemv_q8<2, 2, 256, PRO, EPI, MAP>(ctx, a, name, bytes);
    if (K16 <= 1024) return launch_gemv_q8<1, 2, 1024, PRO, EPI, MAP>(ctx, a, name, bytes);
    if (K16 <= 2048) return launch_gemv_q8<2, 2, 1024, PRO, EPI, MAP>(ctx, a, name, bytes);
    LH_FAIL(ctx, LH_EUNSUPPORTED, "gemv_q8 %s: K=%u exceeds the supported 32768 columns", name, a.K);
}

template <int PRO, int EPI, int MAP>
static int gemv_f32(lh_ctx* ctx, const GemvArgs& a, const char* name);

template <int PRO, int EPI, int MAP>
static int gemv(lh_ctx* ctx, const GemvArgs& a, const char* name, int wtype = 0) {
    return wtype == 7 ? gemv_q8<PRO, EPI, MAP>(ctx, a, name) : gemv_f32<PRO, EPI, MAP>(ctx, a, name);
}

template <int PRO, int EPI, int MAP>
static int gemv_f32(lh_ctx* ctx, const GemvArgs& a, const char* name) {
    // rows are dealt to the workgroups in pairs (the last one takes an odd remainder); only the epilogues that combine two rows
    // (RoPE pairs, w1|w3 interleave) need an even count
    if (a.K % 4 || ((EPI == EPI_QKV_ROPE || EPI == EPI_SILU_MUL) && a.M % 2))
        LH_FAIL(ctx, LH_ESHAPE, "gemv %s: K=%u must be a multiple of 4%s", name, a.K, a.M % 2 ? " and the row count even" : "");
    const uint32_t K4 = a.K / 4;
    const uint64_t bytes = (uint64_t)a.M * a.K * 4;
    const uint32_t rows_wg = a.M / (uint32_t)ctx->ds->num_cu + 4;  // one finishing thread per row, rows * waves partial sums in LDS
    // Workgroup size and rows in flight per wave (U), same-box A/B on the decode loops (tok/s).  7B, 1024 threads:
    // U = 4 for the K = 4096 kernels 219.0, U = 2 222.8, (U at K = 4096, U at K = 11008) = (1,2) 200.0, (3,2) 220.8, (2,3) 222.4,
    // (2,1) 224.1; 512 threads (8 waves, half the per-row reductions and LDS partials), two float4 per thread at K = 4096, six at
    // K = 11008: (2,1) 226.7, (3,1) 224.5, (4,1) 223.4, (2,2) 226.1, (1,1) 209.8; 256 threads at K = 4096 (four float4 per thread): U = 2 227.7,
    // U = 3 227.0.  About 32-48 KB in flight per CU is the sweet
    // spot.  13B, 1024 threads, K = 5120: U = 4 118.8, U = 2 121.4, U = 1 120.6.
    if (K4 <= 4 * 256 && rows_wg <= 256) {  // K <= 4096: 4 waves (K = 4096: 226.0 -> 227.7 tok/s against 8 waves; no gain at K = 11008)
        switch ((K4 + 255) / 256) {
            case 1: return launch_gemv<1, 4, PRO, EPI, MAP, 256>(ctx, a, name, bytes);
            case 2: return launch_gemv<2, 4, PRO, EPI, MAP, 256>(ctx, a, name, bytes);
            case 3: return launch_gemv<3, 2, PRO, EPI, MAP, 256>(ctx, a, name, bytes);
            default: return launch_gemv<4, 2, PRO, EPI, MAP, 256>(ctx, a, name, bytes);
        }
    }
    if (K4 <= 6 * 512 && rows_wg <= 512) {
        switch ((K4 + 511) / 512) {
            case 1: return launch_gemv<1, 2, PRO, EPI, MAP, 512>(ctx, a, name, bytes);
            case 2: return launch_gemv<2, 2, PRO, EPI, MAP, 512>(ctx, a, name, bytes);
            case 3: return launch_gemv<3, 2, PRO, EPI, MAP, 512>(ctx, a, name, bytes);
            case 4: return launch_gemv<4, 1, PRO, EPI, MAP, 512>(ctx, a, name, bytes);
            case 5: return launch_gemv<5, 1, PRO, EPI, MAP, 512>(ctx, a, name, bytes);
            default: return launch_gemv<6, 1, PRO, EPI, MAP, 512>(ctx, a, name, bytes);
        }
    }
    if (rows_wg > (uint32_t)TH) LH_FAIL(ctx, LH_EUNSUPPORTED, "gemv %s: M=%u exceeds %d rows per workgroup", name, a.M, TH - 4);
    switch ((K4 + TH - 1) / TH) {
        case 1: return launch_gemv<1, 2, PRO, EPI, MAP>(ctx, a, name, bytes);
        case 2: return launch_gemv<2, 2, PRO, EPI, MAP>(ctx, a, name, bytes);
        case 3: return launch_gemv<3, 1, PRO, EPI, MAP>(ctx, a, name, bytes);
        case 4: return launch_gemv<4, 2, PRO, EPI, MAP>(ctx, a, name, bytes);
        case 5: return launch_gemv<5, 2, PRO, EPI, MAP>(ctx, a, name, bytes);
        // (six float4 of x, gamma and two rows each do not fit the 128 registers of a 1024-thread workgroup next to the norm: 20 B of scratch)
        case 6: return launch_gemv<6, (PRO == PRO_RMSNORM ? 1 : 2), PRO, EPI, MAP>(ctx, a, name, bytes);
        default: LH_FAIL(ctx, LH_EUNSUPPORTED, "gemv %s: K=%u exceeds the supported 24576 columns", name, a.K);
    }
}

// ---- 2..4 activation rows on the decode weight stream (kernels_rows.h): the same workgroup size and float4-per-thread count as the
// single-row launch of the shape (gemv_f32), so every row's sums are bit-identical to its solo decode step
template <int KI, int U, int THR, int NC, int PRO, int EPI, int MAP>
static int launch_gemv_rows(lh_ctx* ctx, const GemvRowsArgs& a, const char* name, uint64_t bytes) {
    static bool flags[16] = {};
    int rc = set_lds_once(ctx, k_gemv_rows<KI, U, THR, NC, PRO, EPI, MAP>, FAT_LDS, flags);
    if (rc) return rc;
    if (g_prepare_only) return 0;
    ProfScope ps(ctx->stream, name, bytes);
    GemvRowsArgs b = a;
    b.wg_q = (a.M / 2) / (uint32_t)ctx->ds->num_cu; b.wg_r = (a.M / 2) % (uint32_t)ctx->ds->num_cu;   // wg_row_block
    LH_LAUNCH((k_gemv_rows<KI, U, THR, NC, PRO, EPI, MAP>), dim3(ctx->ds->num_cu), dim3(THR), FAT_LDS, ctx->stream, b);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}
// nc = activation rows of the instantiation (2, 4, 8), norm = the launch folds the RMSNorm prologue.  Eight rows keep 8 KI float4 of
// activations per thread: they fit the 512 registers of a 256-thread workgroup's waves (K <= 4096) and, without the norm's gamma registers,
// the 256 of a 512-thread one; the other combinations would spill (ISA-checked: tools/isa_check.py) and are not built.
static bool gemv_rows_shape_ok(lh_ctx* ctx, uint32_t M, uint32_t K, uint32_t nc = 4, bool norm = false) {
    const uint32_t K4 = K / 4, rows_wg = M / (uint32_t)ctx->ds->num_cu + 4;
    if (K % 4 || M % 2) return false;
    if (K4 <= 4 * 256 && rows_wg <= 256) return true;
    if (nc > 4 && (norm || K4 > 6 * 512)) return false;
    return K4 <= 6 * 512 && rows_wg <= 512;
}
template <int NC, int PRO, int EPI, int MAP>
static int gemv_rows_nc(lh_ctx* ctx, const GemvRowsArgs& a, const char* name) {
    if (!gemv_rows_shape_ok(ctx, a.M, a.K, NC, PRO == PRO_RMSNORM)) LH_FAIL(ctx, LH_ESHAPE, "gemv_rows %s: %u x %u has no %d-row instantiation", name, a.M, a.K, NC);
    const uint32_t K4 = a.K / 4, rows_wg = a.M / (uint32_t)ctx->ds->num_cu + 4;
    const uint64_t bytes = (uint64_t)a.M * a.K * 4;
    if (K4 <= 4 * 256 && rows_wg <= 256) {
        // (eight activation rows: three / four weight rows in flight per wave measured slower than two - 5.17 / 5.40 against 4.99 ms per 8-pod tick,
        // profiles/r04_rows_kernel_u.txt)
        switch ((K4 + 255) / 256) {
            case 1: return launch_gemv_rows<1, 4, 256, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
            case 2: return launch_gemv_rows<2, 4, 256, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
            case 3: return launch_gemv_rows<3, 2, 256, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
            default: return launch_gemv_rows<4, 2, 256, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
        }
    }
    if constexpr (NC > 4 && PRO == PRO_RMSNORM) return LH_ESHAPE;   // (refused above)
    else {
    switch ((K4 + 511) / 512) {
        case 1: return launch_gemv_rows<1, 2, 512, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
        case 2: return launch_gemv_rows<2, 2, 512, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
        case 3: return launch_gemv_rows<3, 2, 512, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
        case 4: return launch_gemv_rows<4, 1, 512, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
        case 5: return launch_gemv_rows<5, 1, 512, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
        default: return launch_gemv_rows<6, 1, 512, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
    }
    }
}
// block-int8 twin (k_gemv_q8_rows): the launch shape of gemv_q8's 256-thread workgroups (one or three 16-quant chunks per thread).  (512 threads =
// two row groups per workgroup, two waves per SIMD: measured, nothing - 4 pods 1417 / 1333 against 1393 / 1439 tok/s, 2 pods 939 against 893;
// the kernel issues instructions 73 % of its wave cycles, it does not wait: profiles/r04_q8_rows_threads_pmc.txt)
template <int KI, int U, int NC, int PRO, int EPI, int MAP>
static int launch_gemv_q8_rows(lh_ctx* ctx, const GemvRowsArgs& a, const char* name, uint64_t bytes) {
    static bool flags[16] = {};
    int rc = set_lds_once(ctx, k_gemv_q8_rows<KI, U, 256, 256, NC, PRO, EPI, MAP>, FAT_LDS, flags);
    if (rc) return rc;
    if (g_prepare_only) return 0;
    ProfScope ps(ctx->stream, name, bytes);
    GemvRowsArgs b = a;
    b.wg_q = (a.M / 2) / (uint32_t)ctx->ds->num_cu; b.wg_r = (a.M / 2) % (uint32_t)ctx->ds->num_cu;   // wg_row_block
    LH_LAUNCH((k_gemv_q8_rows<KI, U, 256, 256, NC, PRO, EPI, MAP>), dim3(ctx->ds->num_cu), dim3(256), FAT_LDS, ctx->stream, b);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}
static bool gemv_q8_rows_shape_ok(lh_ctx* ctx, uint32_t M, uint32_t K) {
    const uint32_t K16 = K / 16;
    return K % 32 == 0 && M % 2 == 0 && (K16 <= 256 || (K16 > 512 && K16 <= 768)) && (uint64_t)M / ctx->ds->num_cu + 4 <= 250;
}
template <int NC, int PRO, int EPI, int MAP>
static int gemv_q8_rows_nc(lh_ctx* ctx, const GemvRowsArgs& a, const char* name) {
    if (!gemv_q8_rows_shape_ok(ctx, a.M, a.K)) LH_FAIL(ctx, LH_ESHAPE, "gemv_q8_rows %s: %u x %u has no instantiation", name, a.M, a.K);
    const uint64_t bytes = (uint64_t)a.M * a.K / 32 * 36;
    constexpr int U = NC <= 2 ? 4 : 2;   // rows in flight per register set: the activation rows take the registers the deeper ring would
    if (a.K / 16 <= 256) return launch_gemv_q8_rows<1, U, NC, PRO, EPI, MAP>(ctx, a, name, bytes);
    return launch_gemv_q8_rows<3, (NC <= 2 ? 2 : 1), NC, PRO, EPI, MAP>(ctx, a, name, bytes);
}
template <int PRO, int EPI, int MAP>
static int gemv_rows(lh_ctx* ctx, const GemvRowsArgs& a, const char* name, int wtype = 0) {
    if (wtype == 7) return a.n <= 2 ? gemv_q8_rows_nc<2, PRO, EPI, MAP>(ctx, a, name) : gemv_q8_rows_nc<4, PRO, EPI, MAP>(ctx, a, name);
    if (a.n > 4) return gemv_rows_nc<8, PRO, EPI, MAP>(ctx, a, name);
    return a.n <= 2 ? gemv_rows_nc<2, PRO, EPI, MAP>(ctx, a, name) : gemv_rows_nc<4, PRO, EPI, MAP>(ctx, a, name);
}
// rows the decode stream carries: fp32 up to 8 (round 4; 32 FMAs per 16-byte load = a sixth of a CU's vector rate at the stream's pace),
// block-int8 up to 4 (its 16 converts + 16 NC FMAs per load saturate the vector ALU from there on)
static constexpr uint32_t GEMV_ROWS_MAX_F32 = 8, GEMV_ROWS_MAX_Q8 = 4;
static uint32_t gemv_rows_max(int wtype) { return wtype == 7 ? GEMV_ROWS_MAX_Q8 : GEMV_ROWS_MAX_F32; }
// every launch of a layer (and the lm_head on the last stage) has an n-row instantiation
static bool rows_path_ok(lh_ctx* ctx, const ModelDesc& m, uint32_t n) {
    if (n > gemv_rows_max(m.wtype)) return false;
    const uint32_t nc = n <= 2 ? 2 : (n <= 4 ? 4 : 8);
    auto ok = [&](uint32_t M, uint32_t K, bool norm) { return m.wtype == 7 ? gemv_q8_rows_shape_ok(ctx, M, K) : (m.wtype == 0 && gemv_rows_shape_ok(ctx, M, K, nc, norm)); };
    return m.hd % 2 == 0 && m.d % 4 == 0 && ok(3 * m.d, m.d, true) && ok(m.d, m.d, false) && ok(2 * m.F, m.d, true) && ok(m.d, m.F, false) && (!m.last_stage() || ok(m.V, m.d, true));
}

template <int KI, int U, int NC>
static int launch_cols(lh_ctx* ctx, const GemmColsArgs& a, const char* name) {
    auto kern = k_gemv_cols<KI, U, TH, NC>;
    static bool flags[16] = {};
    int rc = set_lds_once(ctx, kern, FAT_LDS, flags);
    if (rc) return rc;
    if (g_prepare_only) return 0;
    ProfScope ps(ctx->stream, name, (uint64_t)a.M * a.K * 4);
    LH_LAUNCH_AS("k_gemv_cols", kern, dim3(ctx->ds->num_cu), dim3(TH), FAT_LDS, ctx->stream, a);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// LDS-DMA kernel when every tile row starts on a 16-byte boundary (the DMA moves 16-byte granules), register-staged one otherwise
static bool gemm_dma_ok(const GemmArgs& a) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
    const uint32_t ldw = a.ldw ? a.ldw : a.K;
    if (a.K % GBK || a.ldx % 4 || ldw % 4 || a.xbs % 4 || a.wbs % 4 || !al(a.x)) return false;
    for (uint32_t g = 0; g < a.groups; ++g)
        if (!al(a.w[g])) return false;
    return true;
}

// Split-K for launches whose tiles do not fill the CUs evenly (prompts of <= 128 tokens are ONE row of tiles): S ranges of the contraction
// per tile, partial products summed by a second pass.  Cost model in microseconds: rounds of the busiest CU x (slabs per item + ~8 of
// prologue / epilogue) x the time of one 32-deep slab of a bn x bm tile at the CU's fp32 matrix peak (157.3 TFLOP/s / 256), plus the
// reduce pass ((S + 1) x output floats through HBM at ~4 TB/s + a launch).  S must divide the slab count and leave >= 16 slabs per item;
// a split has to win 5 %.  Not only for tiles < #CU: 344 tiles of w1|w3 at 128 rows are two rounds at 67 % occupancy, 688 half-tiles three
// rounds at 90 %.
static uint32_t pick_splitk(uint32_t tiles, uint32_t ncu, uint32_t nkf, uint32_t bn, uint32_t bm, uint64_t out_floats, double* cost_us) {
    const double slab_us = 2.0 * bn * bm * GBK / (157.3e6 / 256.0);
    double best = (double)((tiles + ncu - 1) / ncu) * (nkf + 8) * slab_us;
    uint32_t splits = 1;
    for (uint32_t s2 = 2; s2 <= 32; s2 *= 2) {
        if (nkf % s2 || nkf / s2 < 16) break;
        const double c = (double)(((uint64_t)tiles * s2 + ncu - 1) / ncu) * (nkf / s2 + 8) * slab_us + (double)(s2 + 1) * out_floats * 4.0 / 4e6 + 5.0;
        if (c < best * 0.95) { best = c; splits = s2; }
    }
    if (cost_us) *cost_us = best;
    return splits;
}

template <int WN, int WM, int TN, int TM>
static int launch_gemm(lh_ctx* ctx, const GemmArgs& a0, const char* name, uint32_t batch = 1) {
    constexpr int BN = WN * TN * 32, BM = WM * TM * 32;
    GemmArgs a = a0;
    // the DMA ring needs a few slabs to pay for its prologue and its GST-1 redundant tail slabs: short contractions (Q.K^T, K = 128)
    // stay on the register-staged kernel
    const bool dma = gemm_dma_ok(a) && a.K >= 16 * GBK;
    const uint32_t tiles = ((a.N + BN - 1) / BN) * ((a.M + BM - 1) / BM) * a.groups;
    static bool flags[2][16] = {};
    ProfScope ps(ctx->stream, name, (uint64_t)a.M * a.K * 4 * a.groups);
    if (dma) {
        // persistent: one workgroup per CU (the LDS request is padded past half a CU's 160 KB so that two never co-reside), each
        // looping over its share of the tiles x batch entries
        auto kern = k_gemm_glds<WN, WM, TN, TM>;
        const size_t lds = std::max<size_t>((size_t)GST * (BN + BM) * 32 * sizeof(float), 82 * 1024);
        int rc = set_lds_once(ctx, kern, lds, flags[1]);
        if (rc) return rc;
        if (g_prepare_only) return 0;
        a.batch = batch;
        const uint32_t ncu = (uint32_t)ctx->ds->num_cu;
        uint32_t splits = 1;
        if (!a.causal && batch == 1 && a.epi == GEMM_EPI_STORE && a.M % 4 == 0 && a.ldy % 4 == 0)
            splits = pick_splitk(tiles, ncu, a.K / GBK, BN, BM, (uint64_t)a.groups * a.N * a.M, nullptr);
        if (splits > 1) {
            const uint64_t need = (uint64_t)a.groups * splits * a.N * a.M;
            if (need > ctx->splitk_floats) {
                LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->splitk) LH_HIP(ctx, hipFree(ctx->splitk));
                ctx->splitk = nullptr; ctx->splitk_floats = 0; ctx->splitk_gen++;
                LH_HIP(ctx, hipMalloc((void**)&ctx->splitk, need * 4));
                ctx->splitk_floats = need;
            }
            a.splits = splits;
            a.part = ctx->splitk;
        }
        const uint64_t work = (uint64_t)tiles * batch * splits;
        LH_LAUNCH_AS("k_gemm_glds", kern, dim3((uint32_t)std::min<uint64_t>(work, (uint64_t)ncu)), dim3(256), lds, ctx->stream, a);
        if (splits > 1) {
            const uint64_t quads = (uint64_t)a.groups * a.N * (a.M / 4);
            LH_LAUNCH(k_splitk_reduce, dim3((uint32_t)std::min<uint64_t>((quads + 255) / 256, 4096)), dim3(256), 0, ctx->stream, a);
        }
    } else {
        auto kern = k_gemm_mfma<WN, WM, TN, TM>;
        const size_t lds = (size_t)2 * GBK * (BN + 1 + BM + 1) * sizeof(float);
        int rc = set_lds_once(ctx, kern, lds, flags[0]);
        if (rc) return rc;
        if (g_prepare_only) return 0;
        LH_LAUNCH_AS("k_gemm_mfma", kern, dim3(tiles, batch), dim3(256), lds, ctx->stream, a);
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// block-int8 weights, prompts beyond 128 rows: k_gemm_q8b3 (kernels_gemm_b9.h) - X split into three bf16 planes by one pass in front, three
// exact-product MFMAs per quant block, 128 x 256 tiles.  Standalone at 1024 rows (profiles/r05_gemm_q8b3_probe.txt): 13B wq|wk|wv 524 us against
// k_gemm_q8's 1546, wo 192 / 728, w1|w3 920 / 2566, w2 477 / 1870; 7B 386 / 860, 139 / 280, 569 / 1637, 350 / 732.  One slab of its tile costs
// 0.53 of a 128 x 128 slab of k_gemm_q8 (1.2 us against 2.27): taken where rounds x that beats k_gemm_q8's own best (split-K included).
static uint16_t* ensure_xs3(lh_ctx* ctx, uint64_t elems) {
    if (elems > ctx->xs3_elems) {
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
        if (ctx->xs3) hipFree(ctx->xs3);
        ctx->xs3 = nullptr; ctx->xs3_elems = 0; ctx->splitk_gen++;   // (captured graphs that hold the old planes' address are stale: same generation counter as the split-K buffer)
        if (hipMalloc((void**)&ctx->xs3, elems * 2) != hipSuccess) return nullptr;
        ctx->xs3_elems = elems;
    }
    return ctx->xs3;
}
static bool gemm_q8b3_ok(const GemmArgs& a) {
    auto al = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
    if (a.N <= 64 || a.K % 128 || a.K < 16 * GBK || a.ldx % 4 || !al(a.x)) return false;
    for (uint32_t g = 0; g < a.groups; ++g)
        if (!al(a.w[g]) || !al(a.ws[g])) return false;
    return true;
}
// cost of the launch in units of (slab of a 128 x 128 k_gemm_q8 tile) x 128, and the split-K factor that gives it: a K range is a whole number
// of scale groups (4 slabs), >= 16 slabs, the ranges of a tile differ by at most one group; the reduce pass as in pick_splitk ((S + 1) x output through HBM + a launch)
static double gemm_q8b3_cost(const GemmArgs& a, uint32_t ncu, uint32_t* splits_out) {
    const uint32_t nkf = a.K / GBK, ngr = nkf / 4;
    const uint64_t tiles3 = (uint64_t)((a.N + 127) / 128) * ((a.M + 255) / 256) * a.groups;
    const double unit = 128 * 0.53, us_per_unit = 2.27 / 128.0;
    double best = (double)((tiles3 + ncu - 1) / ncu) * (nkf + 8) * unit;
    uint32_t sp = 1;
    if (a.M % 4 == 0 && a.ldy % 4 == 0)
        for (uint32_t s2 = 2; s2 <= 16; ++s2) {
            if (ngr / s2 < 4) break;                                  // a range keeps >= 16 slabs
            const uint32_t longest = 4 * ((ngr + s2 - 1) / s2);
            const double c = (double)((tiles3 * s2 + ncu - 1) / ncu) * (longest + 8) * unit + ((double)(s2 + 1) * a.groups * a.N * a.M * 4.0 / 4e6 + 5.0) / us_per_unit;
            if (c < best * 0.95) { best = c; sp = s2; }
        }
    *splits_out = sp;
    return best;
}
static int launch_gemm_q8b3(lh_ctx* ctx, GemmArgs a, const char* name, uint32_t splits) {
    auto kern = k_gemm_q8b3<4>;
    const size_t lds = gemm_q8b3_lds_bytes(4);
    static bool flags[16] = {};
    int rc = set_lds_once(ctx, kern, lds, flags);
    if (rc) return rc;
    // (allocations BEFORE the prepare-only return, as the stream kernels do: a capture pass must find xs3 and the split-K buffer sized - ADVICE r5)
    uint16_t* xs = ensure_xs3(ctx, (uint64_t)3 * a.N * a.K);
    if (!xs) LH_FAIL(ctx, LH_ENOMEM, "%s: planes of %u x %u activations", name, a.N, a.K);
    a.xs = xs; a.xs_plane = (uint64_t)a.N * a.K; a.ldxs = a.K;
    a.splits = splits > 1 ? splits : 0; a.part = nullptr;
    if (splits > 1) {
        const uint64_t need = (uint64_t)a.groups * splits * a.N * a.M;
        if (need > ctx->splitk_floats) {
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->splitk) LH_HIP(ctx, hipFree(ctx->splitk));
            ctx->splitk = nullptr; ctx->splitk_floats = 0; ctx->splitk_gen++;
            LH_HIP(ctx, hipMalloc((void**)&ctx->splitk, need * 4));
            ctx->splitk_floats = need;
        }
        a.part = ctx->splitk;
    }
    if (g_prepare_only) return 0;
    {
        TraceScope ts_(ctx->stream, "split3_rows");
        Split3Args sa = {a.x, xs, a.xs_plane, a.K, a.ldx, a.K};
        LH_LAUNCH(k_split3_rows, dim3(a.N), dim3(256), 0, ctx->stream, sa);
    }
    const uint64_t items = (uint64_t)((a.N + 127) / 128) * ((a.M + 255) / 256) * a.groups * (splits > 1 ? splits : 1);
    ProfScope ps(ctx->stream, name, (uint64_t)a.M * a.K / 32 * 36 * a.groups);
    LH_LAUNCH_AS("k_gemm_q8b3", kern, dim3((uint32_t)std::min<uint64_t>(items, (uint64_t)ctx->ds->num_cu)), dim3(512), lds, ctx->stream, a);
    if (splits > 1) {
        const uint64_t quads = (uint64_t)a.groups * a.N * (a.M / 4);
        LH_LAUNCH(k_splitk_reduce, dim3((uint32_t)std::min<uint64_t>((quads + 255) / 256, 4096)), dim3(256), 0, ctx->stream, a);
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// block-int8 weights, prompts beyond the 64 rows of k_stream_q8b (and shapes it is not built for): dequantising MFMA GEMM (k_gemm_q8), persistent one workgroup per CU
static int gemm_q8_group(lh_ctx* ctx, const float* x, uint32_t ldx, uint32_t groups, const float* const* wq, const float* const* wsc, float* const* y,
                         const float* const* r, uint32_t M, uint32_t K, uint32_t n, uint32_t ldy, const char* name) {
    if (K % GBK || ldx % 4 || ((uintptr_t)x & 15)) LH_FAIL(ctx, LH_ESHAPE, "gemm_q8 %s: K=%u / ldx=%u / X alignment not supported", name, K, ldx);
    GemmArgs a = {};
    a.x = x; a.groups = groups; a.N = n; a.M = M; a.K = K; a.ldx = ldx; a.ldy = ldy;
    for (uint32_t g = 0; g < groups; ++g) { a.w[g] = wq[g]; a.ws[g] = wsc[g]; a.y[g] = y[g]; a.r[g] = r ? r[g] : nullptr; }
    // tile shape by the same balance model as the fp32 GEMM (rounds of the busiest CU x tile width); 64-row tiles up to 64 rows;
    // split-K when the launch has fewer tiles than CUs (see launch_gemm)
    const uint32_t ncu = (uint32_t)ctx->ds->num_cu, tn = (n + 127) / 128;
    auto rounds = [&](uint32_t bm) { return (double)((tn * ((M + bm - 1) / bm) * groups + ncu - 1) / ncu) * bm; };
    const int shape = n <= 64 ? 2 : rounds(160) * 1.03 < rounds(128) ? 1 : 0;
    const uint32_t bm = shape == 1 ? 160 : 128;
    const uint32_t tiles = tn * ((M + bm - 1) / bm) * groups;
    if (tiles < ncu && M % 4 == 0 && ldy % 4 == 0) {
        const uint32_t nkf = K / GBK;
        double best = (double)(nkf + 8);
        for (uint32_t s2 = 2; s2 <= 32; s2 *= 2) {
            if (nkf % s2 || nkf / s2 < 16) break;
            const double c = (double)(((uint64_t)tiles * s2 + ncu - 1) / ncu) * (nkf / s2 + 8);
            if (c < best * 0.95) { best = c; a.splits = s2; }
        }
        if (a.splits > 1) {
            const uint64_t need = (uint64_t)groups * a.splits * n * M;
            if (need > ctx->splitk_floats) {
                LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->splitk) LH_HIP(ctx, hipFree(ctx->splitk));
                ctx->splitk = nullptr; ctx->splitk_floats = 0; ctx->splitk_gen++;
                LH_HIP(ctx, hipMalloc((void**)&ctx->splitk, need * 4));
                ctx->splitk_floats = need;
            }
            a.part = ctx->splitk;
        }
    }
    const uint32_t items = tiles * (a.splits ? a.splits : 1);
    if (gemm_q8b3_ok(a)) {
        const uint32_t nkf = K / GBK, sp = a.splits ? a.splits : 1;
        uint32_t sp3 = 1;
        const double c8 = (double)((items + ncu - 1) / ncu) * (nkf / sp + 8) * bm, c3 = gemm_q8b3_cost(a, ncu, &sp3);
        if (c3 < c8) return launch_gemm_q8b3(ctx, a, name, sp3);
    }
    ProfScope ps(ctx->stream, name, (uint64_t)M * K / 32 * 36 * groups);
    int rc;
    static bool flags[3][16] = {};
    if (shape == 1) {
        auto kern = k_gemm_q8<4, 1, 1, 5>;
        const size_t lds = std::max<size_t>((size_t)2 * (128 + 160) * 32 * sizeof(float), 82 * 1024);
        if ((rc = set_lds_once(ctx, kern, lds, flags[1]))) return rc;
        if (g_prepare_only) return 0;
        LH_LAUNCH_AS("k_gemm_q8", kern, dim3(std::min<uint32_t>(items, ncu)), dim3(256), lds, ctx->stream, a);
    } else if (shape == 2) {
        auto kern = k_gemm_q8<2, 2, 1, 2>;
        const size_t lds = std::max<size_t>((size_t)2 * (64 + 128) * 32 * sizeof(float), 82 * 1024);
        if ((rc = set_lds_once(ctx, kern, lds, flags[2]))) return rc;
        if (g_prepare_only) return 0;
        LH_LAUNCH_AS("k_gemm_q8", kern, dim3(std::min<uint32_t>(items, ncu)), dim3(256), lds, ctx->stream, a);
    } else {
        auto kern = k_gemm_q8<2, 2, 2, 2>;
        const size_t lds = std::max<size_t>((size_t)2 * (128 + 128) * 32 * sizeof(float), 82 * 1024);
        if ((rc = set_lds_once(ctx, kern, lds, flags[0]))) return rc;
        if (g_prepare_only) return 0;
        LH_LAUNCH_AS("k_gemm_q8", kern, dim3(std::min<uint32_t>(items, ncu)), dim3(256), lds, ctx->stream, a);
    }
    if (a.splits > 1) {
        const uint64_t quads = (uint64_t)groups * n * (M / 4);
        LH_LAUNCH(k_splitk_reduce, dim3((uint32_t)std::min<uint64_t>((quads + 255) / 256, 4096)), dim3(256), 0, ctx->stream, a);
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}
static int gemm_q8(lh_ctx* ctx, const float* wq, const float* wsc, const float* x, float* y, const float* resid, uint32_t M, uint32_t K, uint32_t n, uint32_t ldx,
                   uint32_t ldy, const char* name) {
    return gemm_q8_group(ctx, x, ldx, 1, &wq, &wsc, &y, resid ? &resid : nullptr, M, K, n, ldy, name);
}

// Large-N prefill attention as batched MFMA GEMMs over heads, mirroring the reference's own structure (llama.go:300-333):
//   S_h = Q_h K_h^T (full block, like MulMat(K, Q))  ->  scale + causal mask + softmax  ->  V^T copy  ->  O_h = P_h V_h.
static int attention_gemm(Plan* p, const float* q, const float* kc, const float* vc, float* out, uint32_t n, uint32_t past, float scale) {
    lh_ctx* ctx = p->ctx;
    const ModelDesc& m = p->md;
    const uint32_t T = past + n, Tp = (T + 31) & ~31u, H = m.H, hd = m.hd, d = m.d;
    const uint64_t need_s = (uint64_t)H * n * Tp, need_v = (uint64_t)H * hd * Tp;
    if (need_s > p->scores_cap) {
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (p->scores) LH_HIP(ctx, hipFree(p->scores));
        p->scores = nullptr; p->scores_cap = 0;
        LH_HIP(ctx, hipMalloc((void**)&p->scores, need_s * 4));
        p->scores_cap = need_s;
    }
    if (need_v > p->vt_cap) {
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (p->vt) LH_HIP(ctx, hipFree(p->vt));
        p->vt = nullptr; p->vt_cap = 0;
        LH_HIP(ctx, hipMalloc((void**)&p->vt, need_v * 4));
        p->vt_cap = need_v;
    }
    int rc;
    {   // S[h][j][t] = sum_c Q[j][h*hd + c] * K[t][h*hd + c]
        GemmArgs a = {};
        a.x = q; a.w[0] = kc; a.y[0] = p->scores; a.groups = 1; a.N = n; a.M = T; a.K = hd; a.ldx = d; a.ldw = d; a.ldy = Tp;
        a.xbs = hd; a.wbs = hd; a.ybs = (uint64_t)n * Tp;
        a.causal = 1; a.past = past;
        if ((rc = launch_gemm<2, 2, 2, 2>(ctx, a, "attn_qk_gemm", H))) return rc;
    }
    LH_LAUNCH(k_softmax_causal, dim3(n, H), dim3(256), 0, ctx->stream, p->scores, n, Tp, past, scale);
    LH_LAUNCH(k_transpose_v, dim3(Tp / 32, hd / 32, H), dim3(256), 0, ctx->stream, vc, p->vt, T, Tp, d, hd);
    LH_HIP(ctx, hipGetLastError());
    {   // O[j][h*hd + c] = sum_t P[h][j][t] * VT[h][c][t]
        GemmArgs a = {};
        a.x = p->scores; a.w[0] = p->vt; a.y[0] = out; a.groups = 1; a.N = n; a.M = hd; a.K = Tp; a.ldx = Tp; a.ldw = Tp; a.ldy = d;
        a.xbs = (uint64_t)n * Tp; a.wbs = (uint64_t)hd * Tp; a.ybs = hd;
        a.causal = 2; a.past = past;
        if ((rc = launch_gemm<2, 2, 2, 1>(ctx, a, "attn_pv_gemm", H))) return rc;
    }
    return 0;
}

// Single-pass causal prefill attention (kernels_attn.h): one kernel per layer, no score tensor, no V^T copy (+ the combine pass of the
// blocks that were cut by key range).  hd = 128 (every LLaMA size).
static int attention_flash(Plan* p, const float* q, const float* kc, const float* vc, float* out, uint32_t n, uint32_t past, float scale) {
    lh_ctx* ctx = p->ctx;
    const ModelDesc& m = p->md;
    FlashArgs a = {};
    a.q = q; a.k_cache = kc; a.v_cache = vc; a.out = out; a.d = m.d; a.H = m.H; a.n = n; a.past = past; a.scale = scale;
    a.nqb = (n + FA_BQ - 1) / FA_BQ;
    const uint32_t slots = 2u * (uint32_t)ctx->ds->num_cu;   // 64 KiB of LDS each: two workgroups per CU
    FaWork& w = p->fa_work;   // attn_worklist.h
    if (w.n != n || w.past != past) flash_work_list(w, n, past, m.H, slots);
    a.chunk = w.chunk; a.qb_cut = w.qb_cut; a.pmax = w.pmax; a.nwork = w.nwork;
    static_assert(sizeof(a.work) == sizeof(w.work), "work list");
    memcpy(a.work, w.work, sizeof(a.work));
    const bool cut = w.chunk != 0 && w.qb_cut < a.nqb;
    if (cut) {
        const uint64_t need = (uint64_t)m.H * (a.nqb - w.qb_cut) * w.pmax * FA_BQ * FA_PSTRIDE;
        if (need > p->fa_part_cap) {
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (p->fa_part) LH_HIP(ctx, hipFree(p->fa_part));
            p->fa_part = nullptr; p->fa_part_cap = 0;
            LH_HIP(ctx, hipMalloc((void**)&p->fa_part, need * 4));
            p->fa_part_cap = need;
        }
        a.part = p->fa_part;
    }
    static bool flags[16] = {};
    int rc = set_lds_once(ctx, k_attn_flash, FA_LDS_BYTES, flags);
    if (rc) return rc;
    if (g_prepare_only) return 0;
    const uint32_t items = (a.nwork ? a.nwork : a.nqb) * m.H, grid = std::min<uint32_t>(items, slots);
    {
        ProfScope ps(ctx->stream, "attn_flash", (uint64_t)2 * (past + n) * m.d * 4);
        LH_LAUNCH(k_attn_flash, dim3(grid), dim3(FA_TH), FA_LDS_BYTES, ctx->stream, a);
    }
    if (cut) {
        ProfScope ps(ctx->stream, "attn_flash_combine", (uint64_t)m.H * (a.nqb - w.qb_cut) * w.pmax * FA_BQ * FA_PSTRIDE * 4);
        LH_LAUNCH(k_attn_flash_combine, dim3(m.H, a.nqb - w.qb_cut), dim3(256), 0, ctx->stream, a);
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// groups (<= 3) weight matrices of equal shape multiplied with the same X in ONE launch (wq|wk|wv, w1|w3): more tiles per
// launch = less tile-count quantisation.  Tile shape chosen per launch: work of the busiest CU = ceil(tiles / #CU) * tile area.
// ---- short prompts, 9..64 rows: the weight-streaming MFMA kernel (kernels_stream.h) ------------------------------------------------
// up to 48 rows = three column tiles (six row tiles + three column tiles, two images: 152 KB of LDS).  Four column tiles do not fit next
// to six row tiles with 128-column chunks (with the row tiles capped at three they measured 11.4-11.9 ms at 33..64 rows, no better than
// the tile GEMM): 49..64 rows run half-length chunks (KC = 64: 2 x (6 + 4) x 16 x 68 floats = 87 KB), fp32 weights only.
// 65..96 rows (round 3): five / six column tiles on the same half-length chunks (2 x (6 + 6) x 16 x 68 floats = 104 KB; 6 x 6 accumulator tiles =
// 144 registers of the MFMA waves).  The tile GEMM's single row of 128-row tiles cost 17.6 ms at 65 rows against 10.2 ms at 64.
static constexpr uint32_t STREAM_ROWS_BUILT = 128, STREAM_ROWS_Q8 = 64, BATCH_ROWS_MAX = 64;
static constexpr uint32_t B9S_MIN_ROWS = 49, B9S_MAX_ROWS = 64;   // fp32 on k_stream_b9 (see b9s_shape_ok).  Re-checked with the final kernel, same box (profiles/
// r06_stream_b9_model_ab.txt): 17 / 32 pods 6.36 / 6.55 ms per tick on it against 5.89 / 6.11, 33 / 40 / 48 a tie (6.82 / 6.94 / 7.15 vs 6.92 / 6.97 / 7.11),
// prompts of 65..128 tokens as two passes 14.0-15.8 ms against 9.7-14.4 in one pass of k_stream_dma: the range stays 49..64
// fp32 prompts of 129..STREAM_TWO_PASS_MAX tokens: every matrix in TWO passes of the stream kernels (ceil(n / 2) rows each) instead of the tile GEMM, whose
// launches cost what 256 rows cost.  7B, ms per Eval, same box (profiles/r06_two_pass_prompts.txt): 129 / 144 / 160 / 176 / 192 tokens 17.8 / 19.4 / 19.5 / 22.2 / 22.2
// against 23.9 / 24.0 / 24.2 / 24.2 / 24.4 on the tile GEMM; from 200 tokens the two passes lose (25.1 against 24.4; 240: 28.3 against 24.7).
static constexpr uint32_t STREAM_TWO_PASS_MAX = 192;
static int stream_nct(uint32_t n) { return (int)((n + 15) / 16); }
static int stream_kc(uint32_t n) { return n <= 48 ? 128 : 64; }
static constexpr uint32_t stream_max_rows() { return STREAM_ROWS_BUILT; }

template <int MAXT, int NCT, int KC>
static int launch_stream(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    static bool flags[16] = {};
    // k_stream_mm2 (loader waves + MFMA waves, two LDS images); shapes whose two images do not fit have no launch here (round 1's kernel in which
    // every wave loaded and multiplied took them until round 6: no LLaMA shape reached it - tests/test_gpu_zz_routes.py; tools/kernels_stream_mm_r1.h)
    if (a.ws[0]) return ST_NA;   // (block-int8: k_stream_q8b)
    constexpr int KC2 = KC <= 256 ? KC : 256;
    if (stream2_lds_bytes(MAXT, NCT, KC2) > 160 * 1024) return ST_NA;
    const uint32_t grid = a.ksplit > 1 ? (uint32_t)ctx->ds->num_cu / a.ksplit * a.ksplit : (uint32_t)ctx->ds->num_cu;
    const size_t lds = std::max<size_t>(stream2_lds_bytes(MAXT, NCT, KC2), 82 * 1024);   // one workgroup per CU
    int rc = set_lds_once(ctx, k_stream_mm2<MAXT, NCT, KC2>, lds, flags);
    if (rc) return rc;
    if (g_prepare_only) return 0;
    ProfScope ps(ctx->stream, name, (uint64_t)a.groups * a.M * a.K * 4);
    LH_LAUNCH((k_stream_mm2<MAXT, NCT, KC2>), dim3(grid), dim3(2 * ST_TH), lds, ctx->stream, a);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}
// ---- k_stream_dma (kernels_stream.h): fp32 weights, two column tiles on, no folded norm.  Variant = (chunk length, images in the ring,
// operand pipelining), chosen per shape by stream_dma_default_variant (the A/B switches of round 4, LLAMAHIP_STREAM_V / LLAMAHIP_ROWS_MAX, are gone:
// profiles/r04_stream_dma_variants.txt and r04_ttft_round3_paths_same_session.json keep what they measured).
constexpr int dma_nimg_fit(int maxt, int nct, int kc, int cap) {
    int n = (int)(160 * 1024 / ((size_t)(maxt + nct) * 16 * kc * 4));
    return n < cap ? n : cap;
}
// registers of an MFMA wave: the accumulator tiles + two operand sets of one k-block each (the pipeline is k-block by k-block)
constexpr bool dma_pipe_ok(int maxt, int ncw) { return maxt * ncw * 4 + 2 * (maxt + ncw) * 4 <= 210; }
template <int MAXT, int NCT, int KC, int NIMG, bool PIPE, int CS = 1>
static int launch_stream_dma(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    static_assert(NIMG >= 2, "ring");
    static bool flags[16] = {};
    auto kern = k_stream_dma<MAXT, NCT, KC, NIMG, PIPE, CS>;
    const size_t lds = std::max<size_t>(stream_dma_lds_bytes(MAXT, NCT, KC, NIMG), 82 * 1024);   // one workgroup per CU
    int rc = set_lds_once(ctx, kern, lds, flags);
    if (rc) return rc;
    if (g_prepare_only) return 0;
    const uint32_t grid = a.ksplit > 1 ? (uint32_t)ctx->ds->num_cu / a.ksplit * a.ksplit : (uint32_t)ctx->ds->num_cu;
    ProfScope ps(ctx->stream, name, (uint64_t)a.groups * a.M * a.K * 4);
    LH_LAUNCH_AS("k_stream_dma", kern, dim3(grid), dim3(2 * ST_TH), lds, ctx->stream, a);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}
// variants: 0 = 64-column chunks, as many images as fit (<= 4); 2 = 128-column chunks, two images; 1 / 3 = the same with pipelined operands
// (measured round 4, profiles/r04_stream_dma_variants.txt: the operand pipeline buys nothing on two to six column tiles - the LDS
// latency behind a barrier is not what idles the matrix pipe - so only the eight-column-tile launches are built with it)
template <int MAXT, int NCT>
static int launch_stream_dma_v(lh_ctx* ctx, const StreamArgs& a, const char* name, int v) {
    constexpr int N64 = dma_nimg_fit(MAXT, NCT, 64, 4), N128 = dma_nimg_fit(MAXT, NCT, 128, 2);
    static_assert(N64 >= 2, "two images of 64-column chunks always fit");
    if constexpr (NCT == 8) {   // eight column tiles: MFMA waves as 2 K-groups x 2 column halves, 64-column chunks
        static_assert(dma_pipe_ok(MAXT, NCT / 2), "column split");
        if (v == 1) return launch_stream_dma<MAXT, NCT, 64, N64, true, 2>(ctx, a, name);
        return launch_stream_dma<MAXT, NCT, 64, N64, false, 2>(ctx, a, name);
    } else if constexpr (NCT == 7) {   // seven (97..112 rows): the four K-groups still hold MAXT x 7 accumulator tiles (168 + 52 operand registers at six row tiles)
        return launch_stream_dma<MAXT, NCT, 64, N64, false>(ctx, a, name);
    } else {
        if constexpr (N128 >= 2) if (v == 2) return launch_stream_dma<MAXT, NCT, 128, 2, false>(ctx, a, name);
        return launch_stream_dma<MAXT, NCT, 64, N64, false>(ctx, a, name);
    }
}
// per shape (7B launches, same box, tools/stream_mm_check mode 4): long chunks where a workgroup streams five or more row tiles next to two
// column tiles (w1|w3 at 17..32 rows: 63.2 us against 67.2-68.4; memory-bound there, and a DMA instruction then moves 512 contiguous bytes
// of a row); everywhere else the deeper ring of short chunks (wq|wk|wv at 32 rows 36.6 against 37.1, wo 19.1 against 25.6, w2 46.5 against 67.2)
static int stream_dma_default_variant(int maxt, int nct) { return (nct == 2 && maxt >= 5) ? 2 : 0; }   // (eight column tiles: the operand pipeline measured 2-5 % slower)
// K-chunk: 128 columns; 256 for single-tile workgroups on long rows (w2: 37.5 -> 34.9 us).  Longer chunks (a whole 1 KB of ONE row per
// load instruction, more bytes in flight) measured slower on the other 7B shapes, and a chunk-major copy of the weights (contiguous
// runs per workgroup) gained 2-13 % at twice the footprint: profiles/r02c_stream_mm_check.txt.
template <int MAXT, int NCT>
static int launch_stream_kc(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    if constexpr (NCT >= 4) return launch_stream<MAXT, NCT, 64>(ctx, a, name);
    else {
        if (MAXT == 1 && a.K > 4096 && a.K % 256 == 0) return launch_stream<MAXT, NCT, (MAXT == 1 ? 256 : 128)>(ctx, a, name);
        return launch_stream<MAXT, NCT, 128>(ctx, a, name);
    }
}
template <int MAXT, int NCT>
static int launch_stream_nct(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    // (one column tile, 9..16 rows, stays on k_stream_mm2 with the norm folded into the launch.  Round 5 built the sixteen-equal-waves LDS-DMA
    // structure of k_stream_q8b for it, norm folded on the operand-read side (tools/kernels_stream_eq.h): correct, and slower on every 7B launch -
    // w1|w3 72.0 us against 66-67, wq|wk|wv 43.3 / 39.5, wo 32.0 / 17.6, w2 79.3 / 37.6 at 16 rows - a workgroup barrier of sixteen waves costs
    // ~0.45 us and fp32 chunks that fit the ring are 64 columns: 64..172 barriers per launch.  profiles/r05_stream_eq_probe.txt)
    if constexpr (NCT >= 2) {
        // (one column tile, 9..16 rows, stays on k_stream_mm2 with the norm folded into the launch: on the LDS-DMA kernel with the norm's own launch
        // in front it measured 6.12-6.14 ms per Eval against 5.63-5.68, 16 pods 5.91 against 5.81 ms per tick - profiles/r04_stream_one_column_tile_probe.txt)
        if (!a.ws[0] && !a.gamma && !a.tiled) return launch_stream_dma_v<MAXT, NCT>(ctx, a, name, stream_dma_default_variant(MAXT, NCT));
    }
    if constexpr (NCT > 6) return ST_NA;   // (k_stream_mm2 holds MAXT x NCT accumulator tiles per wave: up to six column tiles)
    else return launch_stream_kc<MAXT, NCT>(ctx, a, name);
}
template <int MAXT>
static int launch_stream_n(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    if (a.n <= 16) return launch_stream_nct<MAXT, 1>(ctx, a, name);
    if (a.n <= 32) return launch_stream_nct<MAXT, 2>(ctx, a, name);
    if (a.n <= 48) return launch_stream_nct<MAXT, 3>(ctx, a, name);
    if (a.n <= 64) return launch_stream_nct<MAXT, 4>(ctx, a, name);
    if constexpr (MAXT <= 6) {   // (8 x 5 / 8 x 6 accumulator tiles do not fit the registers: those launches take the tile GEMM)
        if (a.n <= 80) return launch_stream_nct<MAXT, 5>(ctx, a, name);
        if (a.n <= 96) return launch_stream_nct<MAXT, 6>(ctx, a, name);
        if (a.n <= 112) return launch_stream_nct<MAXT, 7>(ctx, a, name);   // 97..128 rows (round 4): seven / eight column tiles, fp32 weights only
        return launch_stream_nct<MAXT, 8>(ctx, a, name);
    }
    return ST_NA;
}
static int launch_stream_maxt(lh_ctx* ctx, const StreamArgs& a, const char* name, uint32_t maxt) {
    switch (maxt) {
        case 1: return launch_stream_n<1>(ctx, a, name);
        case 2: return launch_stream_n<2>(ctx, a, name);
        case 3: return launch_stream_n<3>(ctx, a, name);
        case 4: return launch_stream_n<4>(ctx, a, name);
        case 5: case 6: return launch_stream_n<6>(ctx, a, name);
        default: return launch_stream_n<8>(ctx, a, name);
    }
}
// returns ST_NA when the shape is not one the kernel is built for (the caller then takes the tile GEMM)
static int gemm_stream_group(lh_ctx* ctx, const float* x, uint32_t ldx, uint32_t groups, const float* const* w, float* const* y, const float* const* r, uint32_t M,
                             uint32_t K, uint32_t n, uint32_t ldy, const char* name, const StreamArgs* fused = nullptr) {
    if (n > stream_max_rows() || groups > 3 || M % 16 || K % 128 || ldx % 4 || ldy % 4 || ((uintptr_t)x & 15)) return ST_NA;
    const uint32_t ncu = (uint32_t)ctx->ds->num_cu, T = M / 16 * groups;
    // ST_EPI_SILU_MUL deals (w1, w3) tile pairs: twice the pairs' ceiling
    const uint32_t maxt = (fused && fused->epi == ST_EPI_SILU_MUL) ? 2 * ((M / 16 + ncu - 1) / ncu) : (T + ncu - 1) / ncu;
    if (maxt > 8) return ST_NA;
    if (fused && fused->epi != ST_EPI_STORE) {   // the fused epilogues live in the wave-specialised variant only: its two images must fit
        const int mt = maxt <= 4 ? (int)maxt : (maxt <= 6 ? 6 : 8);
        if (stream2_lds_bytes(mt, stream_nct(n), stream_kc(n)) > 160 * 1024) return ST_NA;
        if (fused->epi == ST_EPI_QKV_ROPE && (fused->hd % 4 || M % fused->hd)) return ST_NA;
    }
    StreamArgs a = {};
    if (fused) a = *fused;
    a.x = x; a.groups = groups; a.M = M; a.K = K; a.n = n; a.ldx = ldx; a.ldy = ldy;
    for (uint32_t g = 0; g < groups; ++g) {
        a.w[g] = w[g]; a.y[g] = y ? y[g] : nullptr; a.r[g] = r ? r[g] : nullptr;
        if (((uintptr_t)w[g] & 15) || ((uintptr_t)a.y[g] & 15) || (a.r[g] && ((uintptr_t)a.r[g] & 15))) return ST_NA;
    }
    return launch_stream_maxt(ctx, a, name, maxt);
}

// Single-tile matrices (wo, w2: one 16-row tile per CU) at 17..48 rows: pairs of workgroups split the contraction (StreamArgs::ksplit), so a
// workgroup re-reads half of X out of L2 for twice the rows, and k_stream_reduce_norm adds the two partials + the residual and
// writes the RMSNorm * gamma rows the NEXT matmul reads - it stands where that norm's launch stood.  Standalone, 32 rows (tools/
// stream_mm_check, profiles/r02d_stream_ksplit.txt): w2 55.6 -> 45.6 us, wo 23.9 -> 22.0 us; at 16 rows nothing (37.7 -> 37.0, 17.9 -> 21.1).
// returns ST_NA when not applicable (the caller takes the one-launch path).
static int gemm_stream_split(lh_ctx* ctx, const float* w, const float* x, uint32_t ldx, uint32_t M, uint32_t K, uint32_t n, const float* resid,
                             float* y, const float* gamma, float* h, const char* name) {
    // pairs: four-way splits measured no better, standalone (w2 at 32 / 48 rows: 45.6 / 62.2 us in pairs, 47.6 / 60.6 in fours, + a longer
    // reduce pass) and in the model (40 / 48 tokens: 8.94 / 9.01 ms vs 9.08 / 9.24)
    const uint32_t S = 2u;
    if (n <= 16 || n > stream_max_rows() || M % 16 || M > 8192 || K % 128 || K / 128 < 4 * S || ldx % 4) return ST_NA;
    if ((((uintptr_t)w | (uintptr_t)x | (uintptr_t)y | (uintptr_t)resid | (uintptr_t)gamma | (uintptr_t)h) & 15)) return ST_NA;
    const uint32_t ncu = (uint32_t)ctx->ds->num_cu, ngrp = ncu / S;
    if (ngrp == 0) return ST_NA;
    const uint32_t maxt = (M / 16 + ngrp - 1) / ngrp;
    if (maxt > 8) return ST_NA;
    const int mt = maxt <= 4 ? (int)maxt : (maxt <= 6 ? 6 : 8);
    if (stream2_lds_bytes(mt, stream_nct(n), stream_kc(n)) > 160 * 1024) return ST_NA;
    const uint64_t need = (uint64_t)S * n * M;
    if (need > ctx->splitk_floats) {
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->splitk) LH_HIP(ctx, hipFree(ctx->splitk));
        ctx->splitk = nullptr; ctx->splitk_floats = 0; ctx->splitk_gen++;
        LH_HIP(ctx, hipMalloc((void**)&ctx->splitk, need * 4));
        ctx->splitk_floats = need;
    }
    StreamArgs a = {};
    a.x = x; a.groups = 1; a.M = M; a.K = K; a.n = n; a.ldx = ldx; a.ldy = M; a.w[0] = w; a.y[0] = ctx->splitk;
    a.ksplit = S; a.ysplit = (uint64_t)n * M;
    const int rs = launch_stream_maxt(ctx, a, name, maxt);
    if (rs) return rs;
    if (g_prepare_only) return 0;
    StreamReduceArgs r = {};
    r.part = ctx->splitk; r.stride = a.ysplit; r.resid = resid; r.y = y; r.gamma = gamma; r.h = h; r.S = S; r.d = M; r.ldy = M;
    { TraceScope ts_(ctx->stream, "stream_reduce_norm"); LH_LAUNCH(k_stream_reduce_norm, dim3(n), dim3(256), 0, ctx->stream, r); }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- k_stream_q8b (kernels_stream_q8b.h, round 5): block-int8 weights on the bf16 matrix pipe, activations as three bf16 planes (exact split).
// Chunk length per shape: 256 columns; 512 where a workgroup holds at most three row tiles next to one column tile (wq|wk|wv: half as many
// barriers; standalone on 7B 20.2-22.2 us against 23.5); 128 from three column tiles on and where K is not a multiple of 256.
constexpr int q8b_nimg_fit(int maxt, int xr, int kc) {
    const int n = (int)(160 * 1024 / stream_q8b_image_bytes(maxt, xr, kc));
    return n < 3 ? n : 3;   // (a fourth image measured 1-4 % slower on every 7B launch where it fits, profiles/r05_q8b_probe.txt)
}
template <int MAXT, int NCT, int KC, int XR>
static int launch_stream_q8b(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    constexpr int NIMG = q8b_nimg_fit(MAXT, XR, KC);
    if constexpr (NIMG < 2) return ST_NA;
    else {
        static bool flags[16] = {};
        auto kern = k_stream_q8b<MAXT, NCT, KC, NIMG, XR>;
        const size_t lds = std::max<size_t>((size_t)NIMG * stream_q8b_image_bytes(MAXT, XR, KC), 82 * 1024);   // one workgroup per CU
        int rc = set_lds_once(ctx, kern, lds, flags);
        if (rc) return rc;
        if (g_prepare_only) return 0;
        const uint32_t grid = a.ksplit > 1 ? (uint32_t)ctx->ds->num_cu / a.ksplit * a.ksplit : (uint32_t)ctx->ds->num_cu;
        ProfScope ps(ctx->stream, name, (uint64_t)a.groups * a.M * a.K / 32 * 36);
        LH_LAUNCH_AS("k_stream_q8b", kern, dim3(grid), dim3(Q8B_TH), lds, ctx->stream, a);
        LH_HIP(ctx, hipGetLastError());
        return 0;
    }
}
template <int MAXT, int NCT, int XR>
static int launch_stream_q8b_kc(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    const uint32_t S = a.ksplit > 1 ? a.ksplit : 1u;
    if constexpr (NCT == 1 && MAXT <= 3 && q8b_nimg_fit(MAXT, XR, 512) >= 2) { if (a.K % 512 == 0 && a.K / 512 >= S) return launch_stream_q8b<MAXT, NCT, 512, XR>(ctx, a, name); }
    if constexpr (NCT <= 2 && q8b_nimg_fit(MAXT, XR, 256) >= 2) { if (a.K % 256 == 0 && a.K / 256 >= S) return launch_stream_q8b<MAXT, NCT, 256, XR>(ctx, a, name); }
    if (a.K / 128 < S) return ST_NA;
    return launch_stream_q8b<MAXT, NCT, 128, XR>(ctx, a, name);
}
template <int MAXT>
static int launch_stream_q8b_n(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    if (a.n <= 8) return launch_stream_q8b_kc<MAXT, 1, 8>(ctx, a, name);
    if (a.n <= 16) return launch_stream_q8b_kc<MAXT, 1, 16>(ctx, a, name);
    if (a.n <= 32) return launch_stream_q8b_kc<MAXT, 2, 32>(ctx, a, name);
    if (a.n <= 48) return launch_stream_q8b_kc<MAXT, 3, 48>(ctx, a, name);
    if (a.n <= 64) return launch_stream_q8b_kc<MAXT, 4, 64>(ctx, a, name);
    return ST_NA;
}
static int launch_stream_q8b_maxt(lh_ctx* ctx, const StreamArgs& a, const char* name, uint32_t maxt) {
    switch (maxt) {
        case 1: return launch_stream_q8b_n<1>(ctx, a, name);
        case 2: return launch_stream_q8b_n<2>(ctx, a, name);
        case 3: return launch_stream_q8b_n<3>(ctx, a, name);
        case 4: return launch_stream_q8b_n<4>(ctx, a, name);
        case 5: case 6: return launch_stream_q8b_n<6>(ctx, a, name);
        case 7: case 8: return launch_stream_q8b_n<8>(ctx, a, name);
        default: return ST_NA;
    }
}
// ---- k_stream_b9 (kernels_stream_b9.h, round 6): fp32 weights x the same three bf16 planes, nine exact products per weight on the bf16 matrix pipe.
// Wave-specialised like k_stream_dma (four loader waves, four MFMA waves); up to four column tiles (64 rows), 64-column chunks, as many images
// as fit (<= 4).  Eight row tiles next to four column tiles are not built (the MFMA waves' 4 x 4 accumulator tiles + operands spill).
// Eight of the nine products: xl * wl (<= 2^-32 of its product, far below the fp32 accumulator's own rounding) is dropped - against an f64 product max and
// rms error are unchanged to four digits (tools/b9s_probe, profiles/r06_stream_b9_probe.txt: 7B w1|w3 at 64 rows 1.102e-05 / 6.295e-07 with nine and
// with eight), and the kernel is power-bound (the chip holds 1.6-1.9 GHz under it), so an MFMA saved is time saved: 92.2 -> 88.0 us.  Six products
// (also xm * wl and xl * wm, 2^-24 each) measured 79.8 us at rms 6.311e-07 - not taken: those terms are the size of an fp32 product's own rounding.
static constexpr int B9S_PRODUCTS = 8;
template <int MAXT, int NCT>
static int launch_stream_b9(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    constexpr int NIMG = stream_b9_nimg(MAXT, NCT, 4);
    if constexpr (NIMG < 2 || (MAXT >= 8 && NCT >= 4)) return ST_NA;
    else {
        static bool flags[16] = {};
        auto kern = k_stream_b9<MAXT, NCT, NIMG, B9S_PRODUCTS>;
        const size_t lds = std::max<size_t>((size_t)NIMG * stream_b9_image_bytes(MAXT, NCT), 82 * 1024);   // one workgroup per CU
        int rc = set_lds_once(ctx, kern, lds, flags);
        if (rc) return rc;
        if (g_prepare_only) return 0;
        const uint32_t grid = a.ksplit > 1 ? (uint32_t)ctx->ds->num_cu / a.ksplit * a.ksplit : (uint32_t)ctx->ds->num_cu;
        ProfScope ps(ctx->stream, name, (uint64_t)a.groups * a.M * a.K * 4);
        LH_LAUNCH_AS("k_stream_b9", kern, dim3(grid), dim3(B9S_TH), lds, ctx->stream, a);
        LH_HIP(ctx, hipGetLastError());
        return 0;
    }
}
template <int MAXT>
static int launch_stream_b9_n(lh_ctx* ctx, const StreamArgs& a, const char* name) {
    if (a.K % B9S_KC || a.K / B9S_KC < (a.ksplit > 1 ? a.ksplit : 1u)) return ST_NA;
    if (a.n <= 16) return launch_stream_b9<MAXT, 1>(ctx, a, name);
    if (a.n <= 32) return launch_stream_b9<MAXT, 2>(ctx, a, name);
    if (a.n <= 48) return launch_stream_b9<MAXT, 3>(ctx, a, name);
    if (a.n <= 64) return launch_stream_b9<MAXT, 4>(ctx, a, name);
    return ST_NA;
}
static int launch_stream_b9_maxt(lh_ctx* ctx, const StreamArgs& a, const char* name, uint32_t maxt) {
    switch (maxt) {
        case 1: return launch_stream_b9_n<1>(ctx, a, name);
        case 2: return launch_stream_b9_n<2>(ctx, a, name);
        case 3: return launch_stream_b9_n<3>(ctx, a, name);
        case 4: return launch_stream_b9_n<4>(ctx, a, name);
        case 5: case 6: return launch_stream_b9_n<6>(ctx, a, name);
        case 7: case 8: return launch_stream_b9_n<8>(ctx, a, name);
        default: return ST_NA;
    }
}
static bool stream_b9_built(uint32_t maxt, uint32_t n) { return maxt >= 1 && maxt <= 8 && n >= 1 && n <= 64 && !(maxt > 6 && n > 48); }
// groups (<= 3) matrices of equal shape - block-int8 (wsc: their scale planes) or fp32 (wsc[g] == nullptr: k_stream_b9) - times the same activation
// planes in ONE launch; fused: the epilogue (RoPE + cache append | silu * mul)
static int gemm_q8b_group(lh_ctx* ctx, const uint16_t* xs, uint64_t xs_plane, uint32_t ldxs, uint32_t groups, const float* const* wq, const float* const* wsc, float* const* y,
                          const float* const* r, uint32_t M, uint32_t K, uint32_t n, uint32_t ldy, const char* name, const StreamArgs* fused = nullptr) {
    if (n == 0 || n > STREAM_ROWS_Q8 || groups > 3 || M % 16 || K % 128 || ldxs % 8 || ldy % 4 || ((uintptr_t)xs & 15) || (xs_plane * 2) % 16) return ST_NA;
    const uint32_t ncu = (uint32_t)ctx->ds->num_cu, T = M / 16 * groups;
    const uint32_t maxt = (fused && fused->epi == ST_EPI_SILU_MUL) ? 2 * ((M / 16 + ncu - 1) / ncu) : (T + ncu - 1) / ncu;
    if (maxt > 8) return ST_NA;
    if (fused && fused->epi == ST_EPI_QKV_ROPE && (fused->hd % 4 || M % fused->hd)) return ST_NA;
    StreamArgs a = {};
    if (fused) a = *fused;
    a.xs = xs; a.xs_plane = xs_plane; a.ldxs = ldxs; a.groups = groups; a.M = M; a.K = K; a.n = n; a.ldy = ldy;
    for (uint32_t g = 0; g < groups; ++g) {
        a.w[g] = wq[g]; a.ws[g] = wsc[g]; a.y[g] = y ? y[g] : nullptr; a.r[g] = r ? r[g] : nullptr;
        if (((uintptr_t)wq[g] & 15) || ((uintptr_t)a.y[g] & 15) || (a.r[g] && ((uintptr_t)a.r[g] & 15)) || ((uintptr_t)a.ws[g] & 15)) return ST_NA;
    }
    return a.ws[0] ? launch_stream_q8b_maxt(ctx, a, name, maxt) : launch_stream_b9_maxt(ctx, a, name, maxt);
}
// One matrix as groups of S workgroups that split the contraction (a workgroup then holds S times the rows for a 1 / S of K: the single-tile
// matrices wo and w2 read S times less of X out of L2 and run S times fewer, longer chunk loops), k_stream_reduce_norm adds the S partials +
// the residual in fixed order and writes the RMSNorm * gamma rows of the NEXT matmul as planes (and as fp32 rows when h is given) - it
// stands where that norm's launch stood.  Standalone, 7B (profiles/r05_q8b_probe.txt): wo at 8 / 32 rows 11.5 / 14.5 us in fours against
// 12.5 / - unsplit, w2 17.5 / 26.2 against 20.3 / 39.6.
static int gemm_q8b_split(lh_ctx* ctx, const float* wq, const float* wsc, const uint16_t* xs, uint64_t xs_plane, uint32_t ldxs, uint32_t M, uint32_t K, uint32_t n,
                          const float* resid, float* y, const float* gamma, float* h, uint16_t* hs, uint64_t hs_plane, const char* name) {
    if (n == 0 || n > STREAM_ROWS_Q8 || M % 16 || M > 8192 || M % 4 || K % 128 || ldxs % 8 || ((uintptr_t)xs & 15) || (xs_plane * 2) % 16) return ST_NA;
    if ((((uintptr_t)wq | (uintptr_t)y | (uintptr_t)resid | (uintptr_t)gamma | (uintptr_t)h | (uintptr_t)hs | (uintptr_t)wsc) & 15)) return ST_NA;
    const uint32_t ncu = (uint32_t)ctx->ds->num_cu, nchunks = wsc ? K / (K % 256 == 0 ? 256u : 128u) : K / (uint32_t)B9S_KC;
    uint32_t S = 4;
    while (S > 1 && (nchunks < 2 * S || ncu / S == 0)) S >>= 1;
    const uint32_t ngrp = ncu / S, maxt = (M / 16 + ngrp - 1) / ngrp;
    if (maxt > 8) return ST_NA;
    const uint64_t need = (uint64_t)S * n * M;
    if (need > ctx->splitk_floats) {
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->splitk) LH_HIP(ctx, hipFree(ctx->splitk));
        ctx->splitk = nullptr; ctx->splitk_floats = 0; ctx->splitk_gen++;
        LH_HIP(ctx, hipMalloc((void**)&ctx->splitk, need * 4));
        ctx->splitk_floats = need;
    }
    StreamArgs a = {};
    a.xs = xs; a.xs_plane = xs_plane; a.ldxs = ldxs; a.groups = 1; a.M = M; a.K = K; a.n = n; a.ldy = M; a.w[0] = wq; a.ws[0] = wsc; a.y[0] = ctx->splitk;
    a.ksplit = S; a.ysplit = (uint64_t)n * M;
    const int rs = wsc ? launch_stream_q8b_maxt(ctx, a, name, maxt) : launch_stream_b9_maxt(ctx, a, name, maxt);
    if (rs) return rs;
    if (g_prepare_only) return 0;
    StreamReduceArgs r = {};
    r.part = ctx->splitk; r.stride = a.ysplit; r.resid = resid; r.y = y; r.gamma = gamma; r.h = h; r.S = S; r.d = M; r.ldy = M; r.hs = hs; r.hs_plane = hs_plane;
    { TraceScope ts_(ctx->stream, "stream_reduce_norm"); LH_LAUNCH(k_stream_reduce_norm, dim3(n), dim3(256), 0, ctx->stream, r); }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// fp32 weights, long prompts: the exact bf16 x 9 GEMM (kernels_gemm_b9.h) - 128 x 256 tiles, eight waves, X split into planes by one pass in
// front (n K 10 bytes through HBM: ~1 % of the GEMM it feeds).  Per tile area one slab takes 0.62 (a partly filled round) to 0.73 (every CU
// busy: the chip is then power-bound and clocks down) of k_gemm_glds' time, but the tiles are large: gemm_mfma_group takes it where rounds x slab
// time beats the fp32 kernel's best shape.  13B at 1024 rows inside the model (profiles/r05_p13_kernel_trace.txt): wq|wk|wv 1040 us against
// 1407, w1|w3 2030 against 2380, w2 1100 against 1290-1525, wo 424 against 450 (160 tiles of 128 x 256 against 256 of 128 x 160).
static bool gemm_b9_ok(const GemmArgs& a) {
    return a.N > 128 && !a.causal && a.splits <= 1 && gemm_dma_ok(a) && a.K >= 16 * GBK && a.K % 8 == 0;
}
static int launch_gemm_b9(lh_ctx* ctx, GemmArgs a, const char* name, uint32_t splits) {
    auto kern = k_gemm_b9<1, 8, 4, 1, 2>;
    const size_t lds = 2 * gemm_b9_stage_bytes(128, 256);
    static bool flags[16] = {};
    int rc = set_lds_once(ctx, kern, lds, flags);
    if (rc) return rc;
    if (!ensure_xs3(ctx, (uint64_t)3 * a.N * a.K)) LH_FAIL(ctx, LH_ENOMEM, "%s: planes of %u x %u activations", name, a.N, a.K);
    a.xs = ctx->xs3; a.xs_plane = (uint64_t)a.N * a.K; a.ldxs = a.K;
    a.splits = splits > 1 ? splits : 0; a.part = nullptr;
    if (splits > 1) {
        const uint64_t need = (uint64_t)a.groups * splits * a.N * a.M;
        if (need > ctx->splitk_floats) {
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->splitk) LH_HIP(ctx, hipFree(ctx->splitk));
            ctx->splitk = nullptr; ctx->splitk_floats = 0; ctx->splitk_gen++;
            LH_HIP(ctx, hipMalloc((void**)&ctx->splitk, need * 4));
            ctx->splitk_floats = need;
        }
        a.part = ctx->splitk;
    }
    if (g_prepare_only) return 0;
    {
        TraceScope ts_(ctx->stream, "split3_rows");
        Split3Args sa = {a.x, ctx->xs3, a.xs_plane, a.K, a.ldx, a.K};
        LH_LAUNCH(k_split3_rows, dim3(a.N), dim3(256), 0, ctx->stream, sa);
    }
    const uint64_t items = (uint64_t)((a.N + 127) / 128) * ((a.M + 255) / 256) * a.groups * (splits > 1 ? splits : 1);
    ProfScope ps(ctx->stream, name, (uint64_t)a.M * a.K * 4 * a.groups);
    LH_LAUNCH_AS("k_gemm_b9", kern, dim3((uint32_t)std::min<uint64_t>(items, (uint64_t)ctx->ds->num_cu)), dim3(512), lds, ctx->stream, a);
    if (splits > 1) {
        const uint64_t quads = (uint64_t)a.groups * a.N * (a.M / 4);
        LH_LAUNCH(k_splitk_reduce, dim3((uint32_t)std::min<uint64_t>((quads + 255) / 256, 4096)), dim3(256), 0, ctx->stream, a);
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// fused != nullptr: GEMM_EPI_SILU_MUL (w = {w1, w3}, y[0] = gated output [n][M]) or GEMM_EPI_QKV_ROPE (w = {wq, wk, wv}; outputs in *fused) in
// the epilogue of the LDS-DMA tile GEMM; returns ST_NA when the launch cannot take it (short prompt, split-K, register-staged kernel) and
// the caller runs the plain GEMM + the separate pass.
int gemm_mfma_group(lh_ctx* ctx, const float* x, uint32_t ldx, uint32_t groups, const float* const* w, float* const* y, const float* const* r, uint32_t M,
                    uint32_t K, uint32_t n, uint32_t ldy, const char* name, const GemmArgs* fused = nullptr) {
    if (n <= stream_max_rows() && !fused) {
        const int rs = gemm_stream_group(ctx, x, ldx, groups, w, y, r, M, K, n, ldy, name);
        if (rs != ST_NA) return rs;
    }
    GemmArgs a = {};
    if (fused) a = *fused;
    a.x = x; a.groups = groups; a.N = n; a.M = M; a.K = K; a.ldx = ldx; a.ldy = ldy;
    for (uint32_t g = 0; g < groups; ++g) { a.w[g] = w[g]; a.y[g] = y ? y[g] : nullptr; a.r[g] = r ? r[g] : nullptr; }
    if (fused && fused->epi == GEMM_EPI_SILU_MUL) { a.groups = 1; a.M = 2 * M; }   // one matrix of 2 M virtual rows (w1, w3 interleaved)
    const uint32_t ncu = (uint32_t)ctx->ds->num_cu, tn = (n + 127) / 128;
    // tile width and split-K chosen together: slab times of the busiest CU (pick_splitk) x tile width x the narrow tiles' lower matrix-pipe yield
    const bool may_split = !fused && a.K >= 16 * GBK && gemm_dma_ok(a) && a.M % 4 == 0 && ldy % 4 == 0;
    auto cost = [&](uint32_t bm, double penalty) {
        const uint32_t tiles = tn * ((a.M + bm - 1) / bm) * a.groups;
        double c = (double)((tiles + ncu - 1) / ncu) * (a.K / GBK + 8) * bm;
        if (may_split) pick_splitk(tiles, ncu, a.K / GBK, 128, bm, (uint64_t)a.groups * n * a.M, &c);
        return c * penalty;
    };
    if (fused) {
        // only where the epilogue exists: the DMA kernel, whole contraction per tile (no split-K: at least one tile per CU), long prompts
        if (n <= 64 || !gemm_dma_ok(a) || a.K < 16 * GBK || tn * ((a.M + 159) / 160) * a.groups < ncu) return ST_NA;
        if (fused->epi == GEMM_EPI_QKV_ROPE && (fused->hd % 2 || M % fused->hd)) return ST_NA;
    }
    // up to 64 rows: 64 x 128 tiles (half the matrix work of a 128-row tile whose upper half would be padding)
    if (n <= 64) return launch_gemm<2, 2, 1, 2>(ctx, a, name);
    const double c128 = cost(128, 1.0), c160 = cost(160, 1.03), c64 = cost(64, 1.10);
    if (gemm_b9_ok(a)) {
        const uint64_t tiles9 = (uint64_t)tn * ((a.M + 255) / 256) * a.groups;
        const uint32_t nkf = a.K / GBK;
        const double unit = may_split ? 2.0 * 128 * GBK / (157.3e6 / 256.0) : 1.0;      // cost() is in microseconds when split-K is on the table
        constexpr double B9_SLAB = 0.56;   // a 128 x 256 slab of k_gemm_b9 in units of k_gemm_glds' 128 x 128 slab per tile row (0.62 with nine products, r05; eight: x 8 / 9 measured -9.5 %)
        double c9 = (double)((tiles9 + ncu - 1) / ncu) * (nkf + 8) * 256 * B9_SLAB * unit;
        uint32_t s9 = 1;
        if (may_split)      // (cost in microseconds: the reduce pass as in pick_splitk)
            for (uint32_t s2 = 2; s2 <= 8; ++s2) {
                if (nkf / s2 < 16) break;
                const double c = (double)((tiles9 * s2 + ncu - 1) / ncu) * ((nkf + s2 - 1) / s2 + 8) * 256 * B9_SLAB * unit + (double)(s2 + 1) * a.groups * n * a.M * 4.0 / 4e6 + 5.0;
                if (c < c9 * 0.95) { c9 = c; s9 = s2; }
            }
        if (c9 < 0.97 * std::min(c128, std::min(c160, c64))) return launch_gemm_b9(ctx, a, name, s9);
    }
    if (c160 < c128 && c160 <= c64) return launch_gemm<4, 1, 1, 5>(ctx, a, name);
    if (c64 < c128) return launch_gemm<2, 2, 2, 1>(ctx, a, name);
    return launch_gemm<2, 2, 2, 2>(ctx, a, name);
}

static int gemm_mfma(lh_ctx* ctx, const float* w, const float* x, float* y, const float* resid, uint32_t M, uint32_t K, uint32_t n, uint32_t ldx,
                     uint32_t ldy, const char* name) {
    return gemm_mfma_group(ctx, x, ldx, 1, &w, &y, resid ? &resid : nullptr, M, K, n, ldy, name);
}

// Y[n][M] = X[n][K] . W[M][K]^T (+ resid).  N >= 32: fp32 MFMA GEMM (compute-bound side); smaller N: the weight-streaming
// kernel with NC activation rows in registers (HBM-bound side, weights read once per NC rows).
static constexpr uint32_t MFMA_MIN_ROWS = 9;
// block-int8: below this many tokens single-token steps on the int8 stream (2.1 ms each on 7B) beat the dequantising GEMM
static constexpr uint32_t Q8_GEMM_MIN_ROWS = 9;
int gemm_small_n(lh_ctx* ctx, const float* w, const float* x, float* y, const float* resid, uint32_t M, uint32_t K, uint32_t n,
                 uint32_t ldx, uint32_t ldy, const char* name) {
    if (K % 4) LH_FAIL(ctx, LH_ESHAPE, "gemm %s: K=%u must be a multiple of 4", name, K);
    // from 9 rows on the MFMA GEMM (64-row tiles, split-K) beats two or more passes of the 8-column weight stream (7B, one Eval:
    // 8 rows 10.0 ms and 16 rows 18.4 ms on the stream; 17 rows 11.1 ms on the MFMA path)
    if (n >= MFMA_MIN_ROWS && K % GBK == 0 && ldx % 4 == 0) return gemm_mfma(ctx, w, x, y, resid, M, K, n, ldx, ldy, name);
    if (n >= 2) {   // 2..8 rows on the general path: the streaming MFMA kernel
        const int rs = gemm_stream_group(ctx, x, ldx, 1, &w, &y, resid ? &resid : nullptr, M, K, n, ldy, name);
        if (rs != ST_NA) return rs;
    }
    const uint32_t K4 = K / 4;
    const int ki = (int)((K4 + TH - 1) / TH);
    // activation columns per pass: what fits the 128 registers of a 1024-thread workgroup without scratch (ISA-checked per instantiation)
    const uint32_t NCmax = ki <= 2 ? 8 : (ki <= 4 ? 4 : 2);
    for (uint32_t c0 = 0; c0 < n; c0 += NCmax) {
        GemmColsArgs a;
        a.w = w; a.x = x + (size_t)c0 * ldx; a.y = y + (size_t)c0 * ldy; a.resid = resid ? resid + (size_t)c0 * ldy : nullptr;
        a.M = M; a.K = K; a.ldx = ldx; a.ldy = ldy; a.ncols = std::min(NCmax, n - c0);
        int rc;
        switch (ki) {
            case 1: rc = launch_cols<1, 2, 8>(ctx, a, name); break;
            case 2: rc = launch_cols<2, 2, 8>(ctx, a, name); break;
            case 3: rc = launch_cols<3, 2, 4>(ctx, a, name); break;
            case 4: rc = launch_cols<4, 1, 4>(ctx, a, name); break;
            case 5: rc = launch_cols<5, 1, 2>(ctx, a, name); break;
            case 6: rc = launch_cols<6, 1, 2>(ctx, a, name); break;
            default: LH_FAIL(ctx, LH_EUNSUPPORTED, "gemm %s: K=%u exceeds the supported 24576 columns", name, K);
        }
        if (rc) return rc;
    }
    return 0;
}

// ---- short prompts (2..8 token rows, fp32 weights): one fused pass over the weights per matrix group (kernels_skinny.h) -------
static constexpr uint32_t SKINNY_NP = 8, SKINNY_KC_MAX = 4096;
// shapes the kernel is built for: every contraction length a whole number of ring groups (256 floats), RoPE pairs inside a head
// whole-model check for the streaming MFMA kernel (the 2..8-row prompts prefer it over k_skinny: 6.0 vs 6.5 ms on 7B)
static bool stream_shape_ok(lh_ctx* ctx, const ModelDesc& m) {
    if (m.wtype != 0 || m.d % 128 || m.F % 128) return false;
    const uint32_t ncu = (uint32_t)ctx->ds->num_cu;
    auto maxt = [&](uint32_t rows) { return (rows / 16 + ncu - 1) / ncu; };
    return maxt(3 * m.d) <= 8 && maxt(2 * m.F) <= 8;
}
static bool skinny_ok(const ModelDesc& m, uint32_t n) {
    return m.wtype == 0 && n >= 2 && n <= SKINNY_NP && m.d % SK_GRP == 0 && m.F % SK_GRP == 0 && m.hd % 2 == 0;
}
template <int PRO, int EPI, int MAP>
static int launch_skinny(Plan* p, SkinnyArgs a, const char* name) {
    lh_ctx* ctx = p->ctx;
    auto kern = k_skinny<SKINNY_NP, PRO, EPI, MAP>;
    // K-chunks: the staged activation tile is at most 8 x 4096 floats; chunks are whole ring groups (256 floats) of nearly equal length
    constexpr uint32_t GR = SK_GRP;
    const uint32_t nchunks = (a.K + SKINNY_KC_MAX - 1) / SKINNY_KC_MAX;
    const uint32_t kc_max = ((a.K / GR + nchunks - 1) / nchunks) * GR;
    const uint32_t nwg = (uint32_t)ctx->ds->num_cu, npairs = a.M / 2;
    a.rows_cap = 2 * ((npairs + nwg - 1) / nwg) + 2;
    const size_t lds = skinny_lds_bytes(SKINNY_NP, kc_max, a.rows_cap, a.hd, EPI == EPI_RESID, nchunks > 1, EPI == EPI_QKV_ROPE);
    if (lds > 160 * 1024) LH_FAIL(ctx, LH_EUNSUPPORTED, "skinny %s: %zu bytes of LDS needed", name, lds);
    static bool flags[16] = {};
    int rc0 = set_lds_once(ctx, kern, 160 * 1024, flags);
    if (rc0) return rc0;
    if (g_prepare_only) return 0;
    if (nchunks > 1) {
        const uint64_t need = (uint64_t)SKINNY_NP * a.M;
        if (need > p->part_cap) {
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (p->part) LH_HIP(ctx, hipFree(p->part));
            p->part = nullptr; p->part_cap = 0;
            LH_HIP(ctx, hipMalloc((void**)&p->part, need * 4));
            p->part_cap = need;
        }
    }
    ProfScope ps(ctx->stream, name, (uint64_t)a.M * a.K * 4);
    uint32_t k0 = 0;
    for (uint32_t ch = 0; ch < nchunks; ++ch) {
        const uint32_t kc = std::min(kc_max, a.K - k0);
        SkinnyArgs b = a;
        b.k0 = k0; b.kc = kc;
        b.part_in = ch > 0 ? p->part : nullptr;
        b.part_out = ch + 1 < nchunks ? p->part : nullptr;
        LH_LAUNCH_AS("k_skinny", kern, dim3(nwg), dim3(SK_TH), lds, ctx->stream, b);
        k0 += kc;
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

static int launch_attention(lh_ctx* ctx, const AttnArgs& a, uint32_t max_T) {
    if (a.hd > ATT_TH || ATT_TH % a.hd || a.hd % 4) LH_FAIL(ctx, LH_EUNSUPPORTED, "attention: head dim %u unsupported (needs to divide %d)", a.hd, ATT_TH);
    const size_t lds = (2 * (size_t)((max_T + 63) & ~63u) + ATT_TH) * 4;
    static bool flags[16] = {};
    static size_t cur[16] = {};
    if (lds > 48 * 1024 && lds > cur[ctx->device & 15]) {
        flags[ctx->device & 15] = false;
        int rc = set_lds_once(ctx, k_attention, 160 * 1024, flags);
        if (rc) return rc;
        cur[ctx->device & 15] = 160 * 1024;
    }
    if (lds > 160 * 1024) LH_FAIL(ctx, LH_EUNSUPPORTED, "attention: %u keys exceed the single-pass LDS budget", max_T);
    if (g_prepare_only) return 0;
    if (g_only) return 0;
    ProfScope ps(ctx->stream, "attention", (uint64_t)2 * max_T * a.d * 4);
    LH_LAUNCH(k_attention, dim3(a.d / a.hd, a.n), dim3(ATT_TH), lds, ctx->stream, a);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// decode attention for long-context plans: chunks of ATT_TC keys per workgroup + combine
// (part: [rows][H][chunks][hd + 2] partials; rows = a.n query rows of a batched Eval, 1 otherwise)
static int launch_attention_split(Plan* p, const AttnArgs& a, float* part) {
    lh_ctx* ctx = p->ctx;
    const ModelDesc& m = p->md;
    if (a.hd != 128) LH_FAIL(ctx, LH_EUNSUPPORTED, "split attention: head dim %u", a.hd);
    const uint32_t nch = (m.ctx + ATT_TC - 1) / ATT_TC, nrows = a.rows ? a.n : 1u;
    if (g_prepare_only || g_only) return 0;
    {
        ProfScope ps(ctx->stream, "attention_split", (uint64_t)2 * m.ctx * a.d * 4 * nrows);
        LH_LAUNCH(k_attention_split, dim3(m.H, nch, nrows), dim3(ATT_TH), 0, ctx->stream, a, part);
    }
    {
        ProfScope ps(ctx->stream, "attention_combine", (uint64_t)m.H * nch * (a.hd + 2) * 4 * nrows);
        LH_LAUNCH(k_attention_combine, dim3(m.H, nrows), dim3(128), 0, ctx->stream, a, (const float*)part, nch);
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
static void drop_graphs(Plan* p, uint32_t mask) {
    for (int i = 0; i < Plan::G_COUNT; ++i) {
        if (!(mask & (1u << i))) continue;
        if (p->exec[i]) { hipGraphExecDestroy(p->exec[i]); p->exec[i] = nullptr; }
        if (p->graph[i]) { hipGraphDestroy(p->graph[i]); p->graph[i] = nullptr; }
    }
}
static constexpr uint32_t GM_ADV = (1u << Plan::G_ADV1) | (1u << Plan::G_ADVN), GM_SMP = (1u << Plan::G_SMP1) | (1u << Plan::G_SMPN);

int plan_ensure_rows(Plan* p, uint32_t n) {
    if (n <= p->n_cap) return 0;
    lh_ctx* ctx = p->ctx;
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const ModelDesc& m = p->md;
    auto re = [&](float** ptr, size_t nfloats) -> int {
        if (*ptr) LH_HIP(ctx, hipFree(*ptr));
        *ptr = nullptr;
        LH_HIP(ctx, hipMalloc((void**)ptr, nfloats * 4));
        return 0;
    };
    const size_t nd = (size_t)n * m.d, nf = (size_t)n * m.F;
    int rc = 0;
    rc |= re(&p->xa, nd); rc |= re(&p->xb, nd); rc |= re(&p->h, nd); rc |= re(&p->q, nd); rc |= re(&p->attn, nd); rc |= re(&p->g, nf);
    if (n > 1) { rc |= re(&p->qraw, nd); rc |= re(&p->kraw, nd); rc |= re(&p->vraw, nd); rc |= re(&p->a1, nf); rc |= re(&p->a3, nf); }
    if (m.last_stage()) rc |= re(&p->logits, (size_t)n * m.V);
    if (rc) return LH_EHIP;
    if (n > 1 && (m.wtype == 7 || m.wtype == 0)) {   // the activations as planes: every row of a block-int8 plan, up to k_stream_b9's 64 rows of an fp32 one
        if (p->s3) LH_HIP(ctx, hipFree(p->s3));
        p->s3 = nullptr;
        p->s3_rows = m.wtype == 7 ? n : std::min<uint32_t>(n, B9S_MAX_ROWS);
        LH_HIP(ctx, hipMalloc((void**)&p->s3, (size_t)3 * p->s3_rows * (2 * (size_t)m.d + m.F) * 2));
    }
    if (p->tokens_dev) LH_HIP(ctx, hipFree(p->tokens_dev));
    LH_HIP(ctx, hipMalloc((void**)&p->tokens_dev, (size_t)n * 4));
    p->n_cap = n;
    p->scratch_gen++;
    // captured graphs hold the old scratch addresses
    drop_graphs(p, ~0u);
    return 0;
}

static constexpr uint32_t SP_SLOTS = 64;

int plan_create(lh_ctx* ctx, const ModelDesc& md, Plan** out) {
    if (md.d % md.H || md.d % 4 || md.F % 4) LH_FAIL(ctx, LH_ESHAPE, "plan: embd %u / heads %u / ff %u not supported", md.d, md.H, md.F);
    Plan* p = new Plan();
    p->ctx = ctx;
    p->md = md;
    const char* env = getenv("LLAMAHIP_NO_GRAPH");
    p->use_graph = !(env && env[0] == '1');
    int rc = ensure_rope_table(ctx, md.ctx, md.hd, &p->rope);
    if (rc) { delete p; return rc; }
    hipError_t e = hipMalloc((void**)&p->sp_dev, sizeof(StepParams) * SP_SLOTS);
    if (e == hipSuccess) e = hipHostMalloc((void**)&p->sp_host, sizeof(StepParams) * SP_SLOTS, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&p->argmax_dev, 4);
    if (e != hipSuccess) { set_error(ctx, "plan_create: %s", hipGetErrorString(e)); plan_destroy(p); return LH_ENOMEM; }
    rc = plan_ensure_rows(p, 1);
    if (rc) { plan_destroy(p); return rc; }
    p->hist = md.kv_hist ? md.kv_hist : std::make_shared<std::vector<uint32_t>>();
    if (p->hist->size() < md.ctx) p->hist->resize(md.ctx, Plan::HIST_UNKNOWN);
    if (md.ctx > 256 && md.hd == 128) {   // long contexts: decode attention split over the keys (k_attention_split)
        const size_t nch = (md.ctx + ATT_TC - 1) / ATT_TC;
        e = hipMalloc((void**)&p->attn_part, (size_t)md.H * nch * (md.hd + 2) * 4);
        if (e != hipSuccess) { set_error(ctx, "plan_create: %s", hipGetErrorString(e)); plan_destroy(p); return LH_ENOMEM; }
    }
    *out = p;
    return 0;
}

void plan_destroy(Plan* p) {
    if (!p) return;
    hipSetDevice(p->ctx->device);
    hipStreamSynchronize(p->ctx->stream);
    drop_graphs(p, ~0u);
    if (p->ss_dev) hipFree(p->ss_dev);
    if (p->ring_dev) hipFree(p->ring_dev);
    float* bufs[] = {p->xa, p->xb, p->h, p->qraw, p->kraw, p->vraw, p->q, p->attn, p->a1, p->a3, p->g, p->logits, p->scores, p->vt, p->part, p->fa_part, p->emb};
    for (float* b : bufs) if (b) hipFree(b);
    if (p->tokens_dev) hipFree(p->tokens_dev);
    if (p->sp_dev) hipFree(p->sp_dev);
    if (p->sp_host) hipHostFree(p->sp_host);
    if (p->out_tokens_dev) hipFree(p->out_tokens_dev);
    if (p->argmax_dev) hipFree(p->argmax_dev);
    if (p->attn_part) hipFree(p->attn_part);
    if (p->s3) hipFree(p->s3);
    delete p;
}

Plan* plan_find_or_create(lh_ctx* ctx, const ModelDesc& md, int* rc) {
    for (Plan* p : ctx->plans)
        if (p->md.same(md)) {
            // same addresses, possibly a NEW cache buffer at a recycled address: the history is the buffer's, not the plan's
            if (md.kv_hist && p->hist != md.kv_hist) { p->hist = md.kv_hist; if (p->hist->size() < md.ctx) p->hist->resize(md.ctx, Plan::HIST_UNKNOWN); }
            *rc = 0;
            return p;
        }
    Plan* p = nullptr;
    *rc = plan_create(ctx, md, &p);
    if (*rc) return nullptr;
    ctx->plans.push_back(p);
    return p;
}

void destroy_plans(lh_ctx* ctx) {
    for (Plan* p : ctx->plans) plan_destroy(p);
    ctx->plans.clear();
}

// ---- decode step (N = 1): 5 kernels per layer --------------------------------------------------------
// advance: 0 = logits only, 1 = greedy argmax + loop bookkeeping, 2 = device sampler + loop bookkeeping
static int enqueue_decode(Plan* p, const StepParams* sp, const float* x_in, float* x_out, int argmax_advance, uint32_t* argmax_out,
                          const uint32_t* tokens_dev = nullptr, uint32_t logits_row = 0) {
    lh_ctx* ctx = p->ctx;
    const ModelDesc& m = p->md;
    const double2* rope = p->rope;
    int rc;
    const float* x = p->xa;
    if (m.first_stage()) {
        if (!g_prepare_only && !g_only) {
            ProfScope ps(ctx->stream, "embed", (uint64_t)m.d * 4);
            TraceScope ts_(ctx->stream, "embed1");
            LH_LAUNCH(k_embed, dim3(1), dim3(256), 0, ctx->stream, m.tok_emb, tokens_dev, sp, p->xa, m.d, m.V);
            LH_HIP(ctx, hipGetLastError());
        }
    } else {
        x = x_in;  // residual stream received from the previous stage
    }
    float* xa = p->xa;
    float* xb = p->xb;
    const float scale = (float)(1.0 / sqrt((double)m.d / (double)m.H));  // llama.go:306
    for (uint32_t il = m.layer0; il < m.layer1; ++il) {
        const LayerW& L = m.layers[il];
        const size_t slot = (size_t)(il - m.cache_layer0) * m.ctx * m.d;
        {   // RMSNorm*gamma -> wq|wk|wv -> RoPE(Q, new K) -> K,V appended to the cache   (llama.go:255-297)
            GemvArgs a = {};
            a.w[0] = L.wq; a.w[1] = L.wk; a.w[2] = L.wv; a.ws[0] = L.s_wq; a.ws[1] = L.s_wk; a.ws[2] = L.s_wv; a.rows_per_mat = m.d; a.M = 3 * m.d; a.K = m.d;
            a.x = x; a.gamma = L.attn_norm; a.q_out = p->q; a.k_cache = m.kc + slot; a.v_cache = m.vc + slot;
            a.rope = rope; a.hd = m.hd; a.d = m.d; a.sp = sp;
            if ((rc = gemv<PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK>(ctx, a, "gemv_qkv_rope", m.wtype))) return rc;
        }
        {   // scores, scale, mask, softmax, PV, head merge   (llama.go:300-333)
            AttnArgs a = {};
            a.q = p->q; a.k_cache = m.kc + slot; a.v_cache = m.vc + slot; a.out = p->attn; a.d = m.d; a.hd = m.hd; a.n = 1; a.scale = scale; a.sp = sp;
            if (p->attn_part) { if ((rc = launch_attention_split(p, a, p->attn_part))) return rc; }
            else if ((rc = launch_attention(ctx, a, m.ctx))) return rc;
        }
        {   // wo + residual   (llama.go:336-340)
            GemvArgs a = {};
            a.w[0] = L.wo; a.ws[0] = L.s_wo; a.M = m.d; a.K = m.d; a.x = p->attn; a.resid = x; a.y = xb;
            if ((rc = gemv<PRO_PLAIN, EPI_RESID, MAP_SINGLE>(ctx, a, "gemv_wo_resid", m.wtype))) return rc;
        }
        {   // RMSNorm*gamma -> w1|w3 -> silu(w1 h) * (w3 h)   (llama.go:346-361)
            GemvArgs a = {};
            a.w[0] = L.w1; a.w[1] = L.w3; a.ws[0] = L.s_w1; a.ws[1] = L.s_w3; a.M = 2 * m.F; a.K = m.d; a.x = xb; a.gamma = L.ffn_norm; a.y = p->g;
            if ((rc = gemv<PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>(ctx, a, "gemv_w1w3_silu", m.wtype))) return rc;
        }
        {   // w2 + residual   (llama.go:363-366)
            const bool last = il + 1 == m.layer1;
            GemvArgs a = {};
            a.w[0] = L.w2; a.ws[0] = L.s_w2; a.M = m.d; a.K = m.F; a.x = p->g; a.resid = xb;
            a.y = (last && !m.last_stage()) ? x_out : xa;
            if ((rc = gemv<PRO_PLAIN, EPI_RESID, MAP_SINGLE>(ctx, a, "gemv_w2_resid", m.wtype))) return rc;
        }
        x = xa;
    }
    if (m.last_stage()) {   // final RMSNorm*gamma -> lm_head   (llama.go:374-384)
        {
            GemvArgs a = {};
            a.w[0] = m.output; a.ws[0] = m.s_output; a.M = m.V; a.K = m.d; a.x = x; a.gamma = m.norm; a.y = p->logits + logits_row * (size_t)m.V;
            if ((rc = gemv<PRO_RMSNORM, EPI_STORE, MAP_SINGLE>(ctx, a, "gemv_lmhead", m.wtype))) return rc;
        }
        if (g_only) {
        } else if (argmax_advance == 2 && !g_prepare_only) {
            ProfScope ps(ctx->stream, "sample", (uint64_t)m.V * 4);
            TraceScope ts_(ctx->stream, "sample");
            if ((rc = sample_launch(ctx, p->logits, m.V, p->ss_dev, p->ring_dev, (StepParams*)sp, p->out_tokens_dev, argmax_out, nullptr, nullptr, nullptr, 1, p->smp_topk))) return rc;
        } else if ((argmax_advance || argmax_out) && !g_prepare_only) {
            ProfScope ps(ctx->stream, "argmax", (uint64_t)m.V * 4);
            TraceScope ts_(ctx->stream, "argmax");
            LH_LAUNCH(k_argmax_advance, dim3(1), dim3(1024), 0, ctx->stream, (const float*)p->logits, m.V, (StepParams*)sp, p->out_tokens_dev,
                               argmax_out, argmax_advance ? 1 : 0);
            LH_HIP(ctx, hipGetLastError());
        }
    }
    return 0;
}

static int ensure_out_tokens(Plan* p, uint32_t n) {
    if (n <= p->out_cap) return 0;
    lh_ctx* ctx = p->ctx;
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (p->out_tokens_dev) LH_HIP(ctx, hipFree(p->out_tokens_dev));
    p->out_tokens_dev = nullptr;
    const uint32_t cap = std::max(n, 4096u);
    LH_HIP(ctx, hipMalloc((void**)&p->out_tokens_dev, (size_t)cap * 4));
    p->out_cap = cap;
    // graphs captured the old pointer
    drop_graphs(p, GM_ADV | GM_SMP);
    return 0;
}

// slot: Plan::G_*; the multi-step slots capture GRAPH_MULTI consecutive steps (the step parameters advance in device memory)
static int ensure_decode_graph(Plan* p, int slot) {
    lh_ctx* ctx = p->ctx;
    if (p->exec[slot]) return 0;
    const int adv = slot == Plan::G_STEP ? 0 : (slot == Plan::G_ADV1 || slot == Plan::G_ADVN) ? 1 : 2;
    const uint32_t reps = (slot == Plan::G_ADVN || slot == Plan::G_SMPN) ? Plan::GRAPH_MULTI : 1;
    if (adv) { int rc = ensure_out_tokens(p, 1); if (rc) return rc; }
    g_prepare_only = true;
    int rc = enqueue_decode(p, p->sp_dev, nullptr, nullptr, adv, nullptr);
    g_prepare_only = false;
    if (rc) return rc;
    // Relaxed: other pods' threads keep allocating / copying while this one captures (server.go:88-101 runs pods concurrently).  The
    // captured stream is non-blocking and nothing inside the capture touches the legacy stream; in the stricter modes a
    // synchronous hipMemcpy of ANOTHER thread invalidated this capture (tests/test_gpu_llama.py::test_concurrent_pods_share_one_model).
    LH_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
    for (uint32_t r = 0; r < reps && !rc; ++r) rc = enqueue_decode(p, p->sp_dev, nullptr, nullptr, adv, nullptr);
    hipError_t e = hipStreamEndCapture(ctx->stream, &p->graph[slot]);
    if (rc) return rc;
    if (e != hipSuccess) LH_FAIL(ctx, LH_EHIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    LH_HIP(ctx, hipGraphInstantiate(&p->exec[slot], p->graph[slot], nullptr, nullptr, 0));
    return 0;
}
// n_steps resident steps as graph launches: as many GRAPH_MULTI-step launches as fit, single steps for the rest
static int launch_resident_steps(Plan* p, uint32_t n_steps, int slot1, int slotn) {
    lh_ctx* ctx = p->ctx;
    int rc;
    uint32_t s = 0;
    // both graphs are captured at the first resident call of a context, whatever its length: a short warm-up run then pays for the
    // capture + instantiation of the multi-step graph too, not the first long run after it
    if ((rc = ensure_decode_graph(p, slotn))) return rc;
    if (n_steps >= Plan::GRAPH_MULTI) {
        for (; s + Plan::GRAPH_MULTI <= n_steps; s += Plan::GRAPH_MULTI) LH_HIP(ctx, hipGraphLaunch(p->exec[slotn], ctx->stream));
    }
    if (s < n_steps) {
        if ((rc = ensure_decode_graph(p, slot1))) return rc;
        for (; s < n_steps; ++s) LH_HIP(ctx, hipGraphLaunch(p->exec[slot1], ctx->stream));
    }
    return 0;
}

static int upload_step_params(Plan* p, uint32_t slot, uint32_t token, uint32_t past, uint32_t step) {
    lh_ctx* ctx = p->ctx;
    LH_LAUNCH(k_set_step, dim3(1), dim3(1), 0, ctx->stream, p->sp_dev + slot, token, past, step);
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

int plan_embeddings(Plan* p, uint32_t n, float** out) {
    lh_ctx* ctx = p->ctx;
    const ModelDesc& m = p->md;
    if (!m.last_stage() || n == 0 || n > p->n_cap) LH_FAIL(ctx, LH_EINVAL, "embeddings: the last Eval of this plan had no %u final rows", n);
    if (n > p->emb_cap) {
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (p->emb) LH_HIP(ctx, hipFree(p->emb));
        p->emb = nullptr; p->emb_cap = 0;
        LH_HIP(ctx, hipMalloc((void**)&p->emb, (size_t)n * m.d * 4));
        p->emb_cap = n;
    }
    // every route of plan_eval leaves the residual rows behind the last layer in p->xa; RMSNorm (ml.go:1753-1812) then Mul by the norm weight
    // (ml.go:1877-1914): k_rmsnorm_rows' arithmetic (fp32 squares, f64 sum, one fp32 scale, two roundings per element)
    LH_LAUNCH(k_rmsnorm_rows, dim3(n), dim3(256), 0, ctx->stream, (const float*)p->xa, m.norm, p->emb, m.d);
    LH_HIP(ctx, hipGetLastError());
    *out = p->emb;
    return 0;
}

bool swap_refeed_tokens(const Plan* p, uint32_t past, uint32_t pending, std::vector<uint32_t>* out) {
    out->clear();
    if (p->keep > past) return false;
    const uint32_t n = (past - p->keep) / 2;
    if (n == 0) return true;
    for (uint32_t i = 0; i + 1 < n; ++i) {              // the n - 1 newest evaluated tokens ...
        const uint32_t idx = past - (n - 1) + i;
        const uint32_t t = (p->hist && idx < p->hist->size()) ? (*p->hist)[idx] : Plan::HIST_UNKNOWN;
        if (t == Plan::HIST_UNKNOWN) return false;
        out->push_back(t);
    }
    out->push_back(pending);                            // ... and the pending one, which lastNTokens already holds
    return true;
}

// n_steps resident decode steps (greedy: slots G_ADV*, sampled: G_SMP*) from the state in sp_dev[0] = {token, past, step}, with the reference's
// context swap whenever the window is full (server.go:160-172): the host then learns the ids produced so far (one synchronisation per
// (ctx - keep) / 2 tokens), re-feeds the run as ONE Eval at position keep and goes on.  first_token / past / step0 = what sp_dev[0] holds at
// entry; the ids produced by step i land in out_tokens_dev[step0 + i].  *past_io returns the position behind the last evaluated token.
// first_in_out (step0 >= 1): the first token is entry step0 - 1 of the output list (the sampler's pick behind the prompt) and is read together
// with the ids the steps produce - the caller need not synchronise for it up front (lh_llama_decode_sample paid a host round trip
// per generation for a value that is only needed at a swap or at the end).
static int resident_steps_swapping(Plan* p, uint32_t first_token, uint32_t* past_io, uint32_t step0, uint32_t n_steps, bool sampled, bool first_in_out = false) {
    lh_ctx* ctx = p->ctx;
    const ModelDesc& md = p->md;
    const int slot1 = sampled ? Plan::G_SMP1 : Plan::G_ADV1, slotn = sampled ? Plan::G_SMPN : Plan::G_ADVN;
    uint32_t past = *past_io, done = 0, pending = first_token;
    int rc;
    std::vector<uint32_t> got, refeed;
    // records the tokens evaluated by steps [from, done): step j evaluated (j == 0 ? first_token : the id step j - 1 produced) at position pos0 + (j - from)
    auto learn = [&](uint32_t from, uint32_t pos0) -> int {
        const uint32_t lead = (first_in_out && step0 >= 1) ? 1u : 0u;
        if (done == 0 && !lead) return 0;
        got.resize(done + 1);
        LH_HIP(ctx, hipMemcpyAsync(got.data() + 1 - lead, p->out_tokens_dev + step0 - lead, (size_t)(done + lead) * 4, hipMemcpyDeviceToHost, ctx->stream));
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (lead) { first_token = got[0]; first_in_out = false; }   // (known from here on: later calls read the produced ids only)
        got.erase(got.begin());
        if (done == 0) { pending = first_token; return 0; }
        for (uint32_t j = from; j < done; ++j) p->record(pos0 + (j - from), j == 0 ? first_token : got[j - 1]);
        pending = got[done - 1];
        return 0;
    };
    uint32_t seg_from = 0, seg_pos0 = past;
    while (done < n_steps) {
        if (past >= md.ctx) {
            if ((rc = learn(seg_from, seg_pos0))) return rc;
            if (p->keep >= md.ctx) LH_FAIL(ctx, LH_EINVAL, "context swap: KeepCount %u leaves no room in a window of %u", p->keep, md.ctx);
            if (!swap_refeed_tokens(p, past, pending, &refeed))
                LH_FAIL(ctx, LH_EINVAL, "context swap at position %u: the tokens of the window are not known to this context (evaluate the prompt through it first)", past);
            if (!refeed.empty() && (rc = plan_eval(p, refeed.data(), nullptr, nullptr, (uint32_t)refeed.size(), p->keep, true))) return rc;
            past = p->keep + (uint32_t)refeed.size();
            if ((rc = upload_step_params(p, 0, pending, past, step0 + done))) return rc;
            seg_from = done; seg_pos0 = past;
        }
        const uint32_t k = std::min(n_steps - done, md.ctx - past);
        if (p->use_graph) {
            if ((rc = launch_resident_steps(p, k, slot1, slotn))) return rc;
        } else {
            for (uint32_t s2 = 0; s2 < k; ++s2)
                if ((rc = enqueue_decode(p, p->sp_dev, nullptr, nullptr, sampled ? 2 : 1, nullptr))) return rc;
        }
        done += k; past += k;
    }
    if ((rc = learn(seg_from, seg_pos0))) return rc;   // (also the synchronisation the callers' host copies wait for)
    *past_io = past;
    return 0;
}

int plan_decode_step(Plan* p, uint32_t token, uint32_t past) {
    lh_ctx* ctx = p->ctx;
    const ModelDesc& m = p->md;
    if (past >= m.ctx) LH_FAIL(ctx, LH_EINVAL, "decode: position %u outside the context window of %u", past, m.ctx);
    if (!m.first_stage() || !m.last_stage()) LH_FAIL(ctx, LH_EINVAL, "plan_decode_step needs a whole-model plan");
    p->record(past, token);
    int rc;
    if (p->use_graph) {
        if ((rc = ensure_decode_graph(p, Plan::G_STEP))) return rc;
        if ((rc = upload_step_params(p, 0, token, past, 0))) return rc;
        LH_HIP(ctx, hipGraphLaunch(p->exec[Plan::G_STEP], ctx->stream));
        return 0;
    }
    if ((rc = upload_step_params(p, 0, token, past, 0))) return rc;
    return enqueue_decode(p, p->sp_dev, nullptr, nullptr, false, nullptr);
}

// ---- general Eval on the plan (N >= 1) ----------------------------------------------------------------

// Shapes a batched Eval (rows of different streams, BatchCtx) can take in ONE weight pass: every position-dependent kernel on that route
// reads the row table (k_stream_mm2's RoPE epilogue, k_rope_store, k_attention, k_attention_split).  Routes that do not (k_skinny, the
// tile GEMM's RoPE epilogue above 64 rows, the flash kernel, block-int8 single-token steps) are excluded here; lh_batch then evaluates
// the rows one after the other on their own plans.
// block-int8 on k_stream_q8b: the shapes every launch of an Eval has an instantiation for (kernels_stream_q8b.h; <= 8 row tiles per workgroup
// in each of wq|wk|wv, w1|w3 pairs, the K-split fours of wo / w2 and the output matrix)
static bool q8b_shape_ok(lh_ctx* ctx, const ModelDesc& m) {
    const uint32_t ncu = (uint32_t)ctx->ds->num_cu, d = m.d, F = m.F;
    if (m.wtype != 7 || ncu < 4 || d > 8192 || d % 128 || F % 128 || m.hd % 4 || d % m.hd) return false;
    if ((3 * d / 16 + ncu - 1) / ncu > 8 || 2 * ((F / 16 + ncu - 1) / ncu) > 8 || (d / 16 + ncu / 4 - 1) / (ncu / 4) > 8) return false;
    if (m.last_stage() && (m.V % 16 || (m.V / 16 + ncu - 1) / ncu > 8)) return false;
    return true;
}
// fp32 weights on k_stream_b9 (round 6): the same layer schedule over planes, from four column tiles on (49..64 rows), where k_stream_dma's fp32-input
// MFMAs are the bound.  Inside the model (profiles/r06_pods64_kernel_trace_*.txt): w1|w3 112.7 -> 86.9 us, wo + w2 82 -> 73; at 33..48 rows the
// fp32 kernel is level or ahead (48 pods 6.99 against 7.08 ms per tick), and the standalone probe overstates ITS times by a quarter (random mantissas);
// every launch of the layer must have an instantiation (the output matrix may fall back to the fp32 rows the last reduce pass also writes).
static bool b9s_shape_ok(lh_ctx* ctx, const ModelDesc& m, uint32_t n) {
    const uint32_t ncu = (uint32_t)ctx->ds->num_cu, d = m.d, F = m.F;
    if (m.wtype != 0 || ncu < 4 || n < 2 || n > B9S_MAX_ROWS || d > 8192 || d % 128 || F % 128 || m.hd % 4 || d % m.hd) return false;
    auto split_maxt = [&](uint32_t K) {   // gemm_q8b_split's group size for a contraction of K columns
        uint32_t S = 4;
        while (S > 1 && (K / (uint32_t)B9S_KC < 2 * S || ncu / S == 0)) S >>= 1;
        const uint32_t ngrp = ncu / S;
        return (d / 16 + ngrp - 1) / ngrp;
    };
    return stream_b9_built((3 * d / 16 + ncu - 1) / ncu, n) && stream_b9_built(2 * ((F / 16 + ncu - 1) / ncu), n) && stream_b9_built(split_maxt(d), n) && stream_b9_built(split_maxt(F), n);
}
static constexpr uint32_t Q8B_PROMPT_ROWS = 88;   // a prompt takes up to two 64-row passes; from 89 rows the tile GEMM k_gemm_q8b3 (one 128-row tile: 9.0-9.8 ms flat over
                                                  // 65..128 rows on 7B) is ahead of them (8.6 ms at 65 rows, 8.9 at 80, 9.75 at 96, 11.4 at 128: profiles/r05_gemm_q8b3_probe.txt)
static bool q8_stream_ok(lh_ctx* ctx, const ModelDesc& m, uint32_t n, uint32_t n_min, bool batch_rows = true) {
    return n >= n_min && n <= (batch_rows ? STREAM_ROWS_Q8 : Q8B_PROMPT_ROWS) && q8b_shape_ok(ctx, m);
}
bool plan_batch_rows_ok(const Plan* p, uint32_t n) {
    const ModelDesc& m = p->md;
    if (n < 2 || n > BATCH_ROWS_MAX) return false;
    if (rows_path_ok(p->ctx, m, n)) return true;   // two to eight (block-int8: four) rows ride the decode stream itself
    // block-int8 on the stream kernel: from 3 rows (one pass of the dequantising stream kernel costs 4.9 ms on 7B whatever the row count <= 16, a single-row
    // int8 GEMV step 2.04 ms: two rows are faster one after the other - profiles/r03_pods_one_gpu.jsonl)
    if (m.wtype == 7) return q8_stream_ok(p->ctx, m, n, 3);
    return m.d % GBK == 0 && m.F % GBK == 0;
}

// Block-int8, 5..64 rows (a prompt's tokens or the pods of a tick): the layers on k_stream_q8b.  Per layer 7 launches:
//   [RMSNorm -> planes] (first layer only; afterwards the w2 reduce pass of the layer before writes them)
//   wq|wk|wv + RoPE + cache append | attention (merged heads -> planes) | wo as K-split fours | reduce + residual + RMSNorm -> planes |
//   w1|w3 + silu * mul -> planes | w2 as K-split fours | reduce + residual + the NEXT norm -> planes
// (llama.go:255-366).  Returns ST_NA before anything is enqueued when a launch of the model has no instantiation (the caller's older route
// then takes the Eval).
static int eval_q8b_layers(Plan* p, const float* x, float* x_out_dev, uint32_t n, uint32_t past, bool last_row_only, const BatchCtx* bc) {
    lh_ctx* ctx = p->ctx;
    const ModelDesc& m = p->md;
    const uint32_t d = m.d, F = m.F;
    if (!(m.wtype == 7 ? q8b_shape_ok(ctx, m) : b9s_shape_ok(ctx, m, n))) return ST_NA;
    if (!p->s3 || n > p->s3_rows) return ST_NA;
    const BatchRow* rows = bc ? bc->rows : nullptr;
    const float scale = (float)(1.0 / sqrt((double)m.d / (double)m.H));
    const uint64_t pd = (uint64_t)p->s3_rows * d, pf = (uint64_t)p->s3_rows * F;   // plane strides (elements)
    uint16_t* hs = p->s3;
    uint16_t* as = hs + 3 * pd;
    uint16_t* gs = as + 3 * pd;
    int rc;
#define Q8B_TRY(expr, what) do { if ((rc = (expr))) { if (rc == ST_NA) LH_FAIL(ctx, LH_ESHAPE, "Eval of %u rows over activation planes: no launch for %s (embd %u, ff %u, weight type %u)", n, what, d, F, m.wtype); return rc; } } while (0)
    bool h_ready = false;
    for (uint32_t il = m.layer0; il < m.layer1; ++il) {
        const LayerW& L = m.layers[il];
        const size_t slot = (size_t)(il - m.cache_layer0) * m.ctx * d;
        if (!h_ready) { TraceScope ts_(ctx->stream, "rmsnorm_rows_s3"); if (!g_prepare_only) LH_LAUNCH(k_rmsnorm_rows_s3, dim3(n), dim3(256), 0, ctx->stream, x, L.attn_norm, (float*)nullptr, hs, pd, d); }
        {   // wq|wk|wv -> RoPE(Q, new K rows) -> K, V appended   (llama.go:263-297)
            const float* wqkv[3] = {L.wq, L.wk, L.wv};
            const float* sqkv[3] = {L.s_wq, L.s_wk, L.s_wv};
            StreamArgs fq = {};
            fq.epi = ST_EPI_QKV_ROPE; fq.q_out = p->q; fq.k_cache = m.kc + slot; fq.v_cache = m.vc + slot; fq.rope = p->rope; fq.hd = m.hd; fq.past = past;
            fq.rows = rows; fq.kv_off = slot;
            for (uint32_t b0 = 0; b0 < n; b0 += STREAM_ROWS_Q8) {   // (a prompt of 65..128 tokens: two passes over the weights, still 1.5x faster than the tile GEMM)
                const uint32_t nb = std::min(STREAM_ROWS_Q8, n - b0);
                fq.q_out = p->q + (size_t)b0 * d; fq.past = past + b0;
                Q8B_TRY(gemm_q8b_group(ctx, hs + (size_t)b0 * d, pd, d, 3, wqkv, sqkv, nullptr, nullptr, d, d, nb, d, m.wtype == 7 ? "q8b_wqkv_rope" : "b9s_wqkv_rope", &fq), "wq|wk|wv");
            }
        }
        if (rows) {   // rows of different streams: one query each, against its own cache up to its own position
            AttnArgs a = {};
            a.q = p->q; a.out = p->attn; a.d = d; a.hd = m.hd; a.n = n; a.scale = scale; a.rows = rows; a.kv_off = slot; a.out_s3 = as; a.out_plane = pd;
            if (bc->attn_part) { if ((rc = launch_attention_split(p, a, bc->attn_part))) return rc; }
            else if ((rc = launch_attention(ctx, a, m.ctx))) return rc;
        } else if (n >= 32 && m.hd == FA_HD) {   // a prompt: single pass, online softmax; its rows are split by a pass of their own
            if ((rc = attention_flash(p, p->q, m.kc + slot, m.vc + slot, p->attn, n, past, scale))) return rc;
            Split3Args sa = {p->attn, as, pd, d, d, d};
            if (!g_prepare_only) { TraceScope ts_(ctx->stream, "split3_rows"); LH_LAUNCH(k_split3_rows, dim3(n), dim3(256), 0, ctx->stream, sa); }
        } else if (n >= 32 && m.hd % 32 == 0) {   // other head sizes: batched MFMA GEMMs over heads with a score tensor (as plan_eval's fp32 route does)
            if ((rc = attention_gemm(p, p->q, m.kc + slot, m.vc + slot, p->attn, n, past, scale))) return rc;
            Split3Args sa = {p->attn, as, pd, d, d, d};
            if (!g_prepare_only) { TraceScope ts_(ctx->stream, "split3_rows"); LH_LAUNCH(k_split3_rows, dim3(n), dim3(256), 0, ctx->stream, sa); }
        } else {
            AttnArgs a = {};
            a.q = p->q; a.k_cache = m.kc + slot; a.v_cache = m.vc + slot; a.out = p->attn; a.d = d; a.hd = m.hd; a.n = n; a.scale = scale; a.sp = nullptr; a.past_host = past;
            a.out_s3 = as; a.out_plane = pd;
            if ((rc = launch_attention(ctx, a, past + n))) return rc;
        }
        // wo + residual + RMSNorm * ffn_norm -> planes   (llama.go:336-351)
        for (uint32_t b0 = 0; b0 < n; b0 += STREAM_ROWS_Q8)
            Q8B_TRY(gemm_q8b_split(ctx, L.wo, L.s_wo, as + (size_t)b0 * d, pd, d, d, d, std::min(STREAM_ROWS_Q8, n - b0), x + (size_t)b0 * d, p->xb + (size_t)b0 * d, L.ffn_norm, nullptr,
                                   hs + (size_t)b0 * d, pd, m.wtype == 7 ? "q8b_wo_ksplit" : "b9s_wo_ksplit"), "wo");
        {   // w1|w3 -> silu(w1 h) * (w3 h) -> planes   (llama.go:354-361)
            const float* w13[2] = {L.w1, L.w3};
            const float* s13[2] = {L.s_w1, L.s_w3};
            StreamArgs fa = {};
            fa.epi = ST_EPI_SILU_MUL; fa.ys = gs; fa.ys_plane = pf; fa.ldys = F;
            for (uint32_t b0 = 0; b0 < n; b0 += STREAM_ROWS_Q8) {
                fa.ys = gs + (size_t)b0 * F;
                Q8B_TRY(gemm_q8b_group(ctx, hs + (size_t)b0 * d, pd, d, 2, w13, s13, nullptr, nullptr, F, d, std::min(STREAM_ROWS_Q8, n - b0), F, m.wtype == 7 ? "q8b_w1w3_silu" : "b9s_w1w3_silu", &fa), "w1|w3");
            }
        }
        {   // w2 + residual (+ the next layer's first norm, or the final norm, on the reduce pass)   (llama.go:363-372)
            const bool last = il + 1 == m.layer1;
            float* y = (last && !m.last_stage()) ? x_out_dev : p->xa;
            const float* next_gamma = !last ? m.layers[il + 1].attn_norm : (m.last_stage() ? m.norm : nullptr);
            for (uint32_t b0 = 0; b0 < n; b0 += STREAM_ROWS_Q8)
                Q8B_TRY(gemm_q8b_split(ctx, L.w2, L.s_w2, gs + (size_t)b0 * F, pf, F, d, F, std::min(STREAM_ROWS_Q8, n - b0), p->xb + (size_t)b0 * d, y + (size_t)b0 * d, next_gamma,
                                       (last && m.last_stage()) ? p->h + (size_t)b0 * d : nullptr, next_gamma ? hs + (size_t)b0 * d : nullptr, pd, m.wtype == 7 ? "q8b_w2_ksplit" : "b9s_w2_ksplit"), "w2");
            h_ready = next_gamma != nullptr;
        }
        x = p->xa;
        LH_HIP(ctx, hipGetLastError());
    }
    if (m.last_stage()) {   // lm_head on the rows the caller reads (llama.go:374-384, 394-401); hs / p->h hold RMSNorm * norm of every row
        const uint32_t r0 = last_row_only ? n - 1 : 0, nr = n - r0;
        const float* wv = m.output; const float* sv = m.s_output; float* yv = p->logits + (size_t)r0 * m.V;
        for (uint32_t b0 = 0; b0 < nr; b0 += STREAM_ROWS_Q8) {
            float* yb = yv + (size_t)b0 * m.V;
            const uint32_t nb = std::min(STREAM_ROWS_Q8, nr - b0);
            rc = gemm_q8b_group(ctx, hs + (size_t)(r0 + b0) * d, pd, d, 1, &wv, &sv, &yb, nullptr, m.V, d, nb, m.V, m.wtype == 7 ? "q8b_lmhead" : "b9s_lmhead");
            if (rc == ST_NA && m.wtype == 0 && nb > 1 && m.V % 32 == 0) {
                // fp32 output matrix with eight row tiles per workgroup next to four column tiles (7B at 49..64 rows): no k_stream_b9 launch of that shape - the two halves
                // of the vocabulary have one each (four row tiles).  64 pods, 7B: 172 us on k_stream_dma -> see profiles/r06_lmhead_halves.txt
                const uint32_t Vh = m.V / 2;
                const float* wh[2] = {m.output, m.output + (size_t)Vh * d};
                float* yh[2] = {yb, yb + Vh};
                for (int h2 = 0; h2 < 2; ++h2)
                    if ((rc = gemm_q8b_group(ctx, hs + (size_t)(r0 + b0) * d, pd, d, 1, &wh[h2], &sv, &yh[h2], nullptr, Vh, d, nb, m.V, "b9s_lmhead"))) break;
            }
            if (rc == ST_NA && m.wtype == 0) {   // fp32 output matrix without a k_stream_b9 launch (eight row tiles next to four column tiles): the fp32 rows
                if (nb == 1) {
                    GemvArgs ga = {};
                    ga.w[0] = m.output; ga.M = m.V; ga.K = d; ga.x = p->h + (size_t)(r0 + b0) * d; ga.y = yb;
                    rc = gemv<PRO_PLAIN, EPI_STORE, MAP_SINGLE>(ctx, ga, "gemv_lmhead_row", 0);
                } else rc = gemm_small_n(ctx, m.output, p->h + (size_t)(r0 + b0) * d, yb, nullptr, m.V, d, nb, d, m.V, "gemm_lmhead");
            }
            Q8B_TRY(rc, "lm_head");
        }
    }
    LH_HIP(ctx, hipGetLastError());
#undef Q8B_TRY
    return 0;
}

int plan_eval(Plan* p, const uint32_t* tokens_host, const float* x_in_dev, float* x_out_dev, uint32_t n, uint32_t past, bool last_row_only, const BatchCtx* bc) {
    lh_ctx* ctx = p->ctx;
    const ModelDesc& m = p->md;
    if (n == 0) LH_FAIL(ctx, LH_EINVAL, "Eval: empty token batch");
    if (bc) {
        if (!bc->rows || (m.first_stage() && !bc->tok_dev) || !plan_batch_rows_ok(p, n)) LH_FAIL(ctx, LH_EINVAL, "Eval: batched rows need a row table, token ids and a supported shape");
        if (n > p->n_cap) LH_FAIL(ctx, LH_EINVAL, "Eval: batched rows exceed the plan's scratch (plan_ensure_rows first: a captured graph must not allocate)");
    } else {
        if ((uint64_t)past + n > m.ctx) LH_FAIL(ctx, LH_EINVAL, "Eval: past %u + n %u exceeds the context window of %u", past, n, m.ctx);
        if (m.first_stage() && !tokens_host) LH_FAIL(ctx, LH_EINVAL, "Eval: first stage needs token ids");
        if (m.first_stage())  // Go panics on tokEmbeddings.Data[id*NE[0]:] past the table (ml.go:1748); the GPU must never gather out of range
            for (uint32_t i = 0; i < n; ++i)
                if (tokens_host[i] >= m.V) LH_FAIL(ctx, LH_EINVAL, "Eval: token id %u at index %u outside the vocabulary of %u", tokens_host[i], i, m.V);
        if (m.first_stage()) for (uint32_t i = 0; i < n; ++i) p->record(past + i, tokens_host[i]);
    }
    if (!m.first_stage() && !x_in_dev) LH_FAIL(ctx, LH_EINVAL, "Eval: later stage needs the residual stream");
    if (!m.last_stage() && !x_out_dev) LH_FAIL(ctx, LH_EINVAL, "Eval: non-final stage needs an output buffer");
    int rc;
    if (!bc && (rc = plan_ensure_rows(p, n))) return rc;
    const BatchRow* rows = bc ? bc->rows : nullptr;
    if (n == 1) {
        if (m.first_stage() && m.last_stage() && p->use_graph) return plan_decode_step(p, tokens_host[0], past);
        const uint32_t slot = 1 + (p->slot_counter++ % (SP_SLOTS - 1));
        if ((rc = upload_step_params(p, slot, tokens_host ? tokens_host[0] : 0, past, 0))) return rc;
        return enqueue_decode(p, p->sp_dev + slot, x_in_dev, x_out_dev, false, nullptr);
    }
    if (rows_path_ok(ctx, m, n)) {
        // ---- 2..8 rows (block-int8: 2..4) (a prompt of that many tokens, or a tick of that many pods): the decode launches with NC activation rows each
        // (kernels_rows.h): 5 launches per layer like the decode step, every row bit-identical to its solo step
        const float* x = p->xa;
        if (m.first_stage()) {
            const uint32_t* tok_dev = bc ? bc->tok_dev : p->tokens_dev;
            if (!bc) {
                if ((rc = ensure_staging(ctx, (uint64_t)n * 4))) return rc;
                LH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // staging reuse
                memcpy(ctx->staging, tokens_host, (size_t)n * 4);
                LH_HIP(ctx, hipMemcpyAsync(p->tokens_dev, ctx->staging, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
            }
            if (!g_prepare_only) LH_LAUNCH(k_embed, dim3(n), dim3(256), 0, ctx->stream, m.tok_emb, tok_dev, (const StepParams*)nullptr, p->xa, m.d, m.V);
            LH_HIP(ctx, hipGetLastError());
        } else {
            x = x_in_dev;
        }
        const float scale = (float)(1.0 / sqrt((double)m.d / (double)m.H));
        const uint32_t d = m.d, F = m.F;
        for (uint32_t il = m.layer0; il < m.layer1; ++il) {
            const LayerW& L = m.layers[il];
            const size_t slot = (size_t)(il - m.cache_layer0) * m.ctx * d;
            {   // RMSNorm*gamma -> wq|wk|wv -> RoPE(Q, new K rows) -> K, V appended   (llama.go:255-297)
                GemvRowsArgs a = {};
                a.w[0] = L.wq; a.w[1] = L.wk; a.w[2] = L.wv; a.ws[0] = L.s_wq; a.ws[1] = L.s_wk; a.ws[2] = L.s_wv; a.rows_per_mat = d; a.M = 3 * d; a.K = d; a.x = x; a.ldx = d; a.n = n; a.gamma = L.attn_norm;
                a.q_out = p->q; a.k_cache = m.kc + slot; a.v_cache = m.vc + slot; a.rope = p->rope; a.hd = m.hd; a.d = d; a.past = past; a.rows = rows; a.kv_off = slot;
                if ((rc = gemv_rows<PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK>(ctx, a, "rows_qkv_rope", m.wtype))) return rc;
            }
            {   // scores, scale, mask, softmax, PV, head merge per row   (llama.go:300-333)
                AttnArgs a = {};
                a.q = p->q; a.out = p->attn; a.d = d; a.hd = m.hd; a.n = n; a.scale = scale;
                if (rows) {
                    a.rows = rows; a.kv_off = slot;
                    if (bc->attn_part) { if ((rc = launch_attention_split(p, a, bc->attn_part))) return rc; }
                    else if ((rc = launch_attention(ctx, a, m.ctx))) return rc;
                } else {
                    a.k_cache = m.kc + slot; a.v_cache = m.vc + slot; a.sp = nullptr; a.past_host = past;
                    if ((rc = launch_attention(ctx, a, past + n))) return rc;
                }
            }
            {   // wo + residual   (llama.go:336-340)
                GemvRowsArgs a = {};
                a.w[0] = L.wo; a.ws[0] = L.s_wo; a.M = d; a.K = d; a.x = p->attn; a.ldx = d; a.n = n; a.y = p->xb; a.resid = x; a.ldy = d;
                if ((rc = gemv_rows<PRO_PLAIN, EPI_RESID, MAP_SINGLE>(ctx, a, "rows_wo_resid", m.wtype))) return rc;
            }
            {   // RMSNorm*gamma -> w1|w3 -> silu(w1 h) * (w3 h)   (llama.go:346-361)
                GemvRowsArgs a = {};
                a.w[0] = L.w1; a.w[1] = L.w3; a.ws[0] = L.s_w1; a.ws[1] = L.s_w3; a.M = 2 * F; a.K = d; a.x = p->xb; a.ldx = d; a.n = n; a.gamma = L.ffn_norm; a.y = p->g; a.ldy = F;
                if ((rc = gemv_rows<PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>(ctx, a, "rows_w1w3_silu", m.wtype))) return rc;
            }
            {   // w2 + residual   (llama.go:363-366)
                const bool last = il + 1 == m.layer1;
                GemvRowsArgs a = {};
                a.w[0] = L.w2; a.ws[0] = L.s_w2; a.M = d; a.K = F; a.x = p->g; a.ldx = F; a.n = n; a.resid = p->xb; a.ldy = d;
                a.y = (last && !m.last_stage()) ? x_out_dev : p->xa;
                if ((rc = gemv_rows<PRO_PLAIN, EPI_RESID, MAP_SINGLE>(ctx, a, "rows_w2_resid", m.wtype))) return rc;
            }
            x = p->xa;
        }
        if (m.last_stage()) {   // final RMSNorm*gamma -> lm_head: the rows the caller reads (llama.go:372-384, 394-401)
            const uint32_t r0 = last_row_only ? n - 1 : 0, nr = n - r0;
            if (nr == 1) {
                GemvArgs a = {};
                a.w[0] = m.output; a.ws[0] = m.s_output; a.M = m.V; a.K = d; a.x = x + (size_t)r0 * d; a.gamma = m.norm; a.y = p->logits + (size_t)r0 * m.V;
                if ((rc = gemv<PRO_RMSNORM, EPI_STORE, MAP_SINGLE>(ctx, a, "gemv_lmhead", m.wtype))) return rc;
            } else {
                GemvRowsArgs a = {};
                a.w[0] = m.output; a.ws[0] = m.s_output; a.M = m.V; a.K = d; a.x = x + (size_t)r0 * d; a.ldx = d; a.n = nr; a.gamma = m.norm; a.y = p->logits + (size_t)r0 * m.V; a.ldy = m.V;
                if ((rc = gemv_rows<PRO_RMSNORM, EPI_STORE, MAP_SINGLE>(ctx, a, "rows_lmhead", m.wtype))) return rc;
            }
        }
        LH_HIP(ctx, hipGetLastError());
        return 0;
    }
    // block-int8: 5..64 rows of a tick / 5..88 of a prompt on k_stream_q8b (two 64-row passes from 65); from 89 rows the tile GEMM k_gemm_q8b3 below
    const bool q8_stream = q8_stream_ok(ctx, m, n, 3, bc != nullptr);
    if (m.wtype == 7 && !q8_stream && (n < Q8_GEMM_MIN_ROWS || m.d % GBK || m.F % GBK || m.hd % 32)) {
        // block-int8, short batches: n causal single-token steps on the int8 weight stream (bit-identical to what the decode
        // path produces for them), logits row i from step i like llama.go:384.  n >= 32 takes the dequantising GEMM below.
        // On a pipeline stage row i of the received / forwarded residual stream stands in for the token id / the logits row.
        for (uint32_t i = 0; i < n; ++i) {
            const uint32_t slot = 1 + (p->slot_counter++ % (SP_SLOTS - 1));
            if ((rc = upload_step_params(p, slot, m.first_stage() ? tokens_host[i] : 0, past + i, 0))) return rc;
            if ((rc = enqueue_decode(p, p->sp_dev + slot, m.first_stage() ? nullptr : x_in_dev + (size_t)i * m.d, m.last_stage() ? nullptr : x_out_dev + (size_t)i * m.d, false,
                                     nullptr, nullptr, i)))
                return rc;
        }
        return 0;
    }
    if (!bc && skinny_ok(m, n) && !stream_shape_ok(ctx, m)) {   // k_skinny only where the streaming MFMA kernel is not built for the shape
        // ---- short prompt: 4 fused weight passes per layer + the per-query attention kernel (like the decode step, n rows wide)
        const float* x = p->xa;
        if (m.first_stage()) {
            if ((rc = ensure_staging(ctx, (uint64_t)n * 4))) return rc;
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // staging reuse
            memcpy(ctx->staging, tokens_host, (size_t)n * 4);
            LH_HIP(ctx, hipMemcpyAsync(p->tokens_dev, ctx->staging, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
            LH_LAUNCH(k_embed, dim3(n), dim3(256), 0, ctx->stream, m.tok_emb, (const uint32_t*)p->tokens_dev, (const StepParams*)nullptr, p->xa, m.d, m.V);
            LH_HIP(ctx, hipGetLastError());
        } else {
            x = x_in_dev;
        }
        const float scale = (float)(1.0 / sqrt((double)m.d / (double)m.H));
        const uint32_t d = m.d, F = m.F;
        for (uint32_t il = m.layer0; il < m.layer1; ++il) {
            const LayerW& L = m.layers[il];
            const size_t slot = (size_t)(il - m.cache_layer0) * m.ctx * d;
            {   // RMSNorm*gamma -> wq|wk|wv -> RoPE(Q, new K rows) -> K,V appended   (llama.go:255-297)
                SkinnyArgs a = {};
                a.w[0] = L.wq; a.w[1] = L.wk; a.w[2] = L.wv; a.rows_per_mat = d; a.M = 3 * d; a.K = d; a.x = x; a.ldx = d; a.n = n; a.gamma = L.attn_norm;
                a.q_out = p->q; a.k_cache = m.kc + slot; a.v_cache = m.vc + slot; a.rope = p->rope; a.hd = m.hd; a.d = d; a.past = past;
                if ((rc = launch_skinny<PRO_RMSNORM, EPI_QKV_ROPE, MAP_BLOCK>(p, a, "skinny_qkv_rope"))) return rc;
            }
            {
                AttnArgs a = {};
                a.q = p->q; a.k_cache = m.kc + slot; a.v_cache = m.vc + slot; a.out = p->attn; a.d = d; a.hd = m.hd; a.n = n; a.scale = scale; a.sp = nullptr; a.past_host = past;
                if ((rc = launch_attention(ctx, a, past + n))) return rc;
            }
            {   // wo + residual   (llama.go:336-340)
                SkinnyArgs a = {};
                a.w[0] = L.wo; a.M = d; a.K = d; a.x = p->attn; a.ldx = d; a.n = n; a.y = p->xb; a.resid = x; a.ldy = d;
                if ((rc = launch_skinny<PRO_PLAIN, EPI_RESID, MAP_SINGLE>(p, a, "skinny_wo_resid"))) return rc;
            }
            {   // RMSNorm*gamma -> w1|w3 -> silu(w1 h) * (w3 h)   (llama.go:346-361)
                SkinnyArgs a = {};
                a.w[0] = L.w1; a.w[1] = L.w3; a.M = 2 * F; a.K = d; a.x = p->xb; a.ldx = d; a.n = n; a.gamma = L.ffn_norm; a.y = p->g; a.ldy = F;
                if ((rc = launch_skinny<PRO_RMSNORM, EPI_SILU_MUL, MAP_PAIR>(p, a, "skinny_w1w3_silu"))) return rc;
            }
            {   // w2 + residual   (llama.go:363-366)
                const bool last = il + 1 == m.layer1;
                SkinnyArgs a = {};
                a.w[0] = L.w2; a.M = d; a.K = F; a.x = p->g; a.ldx = F; a.n = n; a.resid = p->xb; a.ldy = d;
                a.y = (last && !m.last_stage()) ? x_out_dev : p->xa;
                if ((rc = launch_skinny<PRO_PLAIN, EPI_RESID, MAP_SINGLE>(p, a, "skinny_w2_resid"))) return rc;
            }
            x = p->xa;
        }
        if (m.last_stage()) {   // final RMSNorm*gamma -> lm_head: the rows the caller reads (llama.go:372-384, 394-401)
            const uint32_t r0 = last_row_only ? n - 1 : 0;
            if (n - r0 == 1) {
                GemvArgs a = {};
                a.w[0] = m.output; a.M = m.V; a.K = d; a.x = x + (size_t)r0 * d; a.gamma = m.norm; a.y = p->logits + (size_t)r0 * m.V;
                if ((rc = gemv<PRO_RMSNORM, EPI_STORE, MAP_SINGLE>(ctx, a, "gemv_lmhead", 0))) return rc;
            } else {
                SkinnyArgs a = {};
                a.w[0] = m.output; a.M = m.V; a.K = d; a.x = x; a.ldx = d; a.n = n; a.gamma = m.norm; a.y = p->logits; a.ldy = m.V;
                if ((rc = launch_skinny<PRO_RMSNORM, EPI_STORE, MAP_SINGLE>(p, a, "skinny_lmhead"))) return rc;
            }
        }
        LH_HIP(ctx, hipGetLastError());
        return 0;
    }
    // ---- prefill, N > 1 rows (or the rows of a batched Eval: `rows` set, every row at its own position of its own cache)
    const double2* rope = p->rope;
    const float* x = p->xa;
    if (m.first_stage()) {
        const uint32_t* tok_dev = bc ? bc->tok_dev : p->tokens_dev;
        if (!bc) {
            if ((rc = ensure_staging(ctx, (uint64_t)n * 4))) return rc;
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // staging reuse
            memcpy(ctx->staging, tokens_host, (size_t)n * 4);
            LH_HIP(ctx, hipMemcpyAsync(p->tokens_dev, ctx->staging, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        }
        TraceScope ts_(ctx->stream, "embedN");
        LH_LAUNCH(k_embed, dim3(n), dim3(256), 0, ctx->stream, m.tok_emb, tok_dev, (const StepParams*)nullptr, p->xa, m.d, m.V);
        LH_HIP(ctx, hipGetLastError());
    } else {
        x = x_in_dev;
    }
    const float scale = (float)(1.0 / sqrt((double)m.d / (double)m.H));
    const uint32_t d = m.d, F = m.F;
    if (q8_stream) {   // block-int8, up to 64 rows: the bf16 matrix pipe (kernels_stream_q8b.h)
        const int rs = eval_q8b_layers(p, x, x_out_dev, n, past, last_row_only, bc);
        if (rs == ST_NA) LH_FAIL(ctx, LH_ESHAPE, "block-int8 Eval of %u rows: the plan's shape has no k_stream_q8b launch", n);
        return rs;
    }
    static const int b9s_min = getenv("LLAMAHIP_B9S_MIN") ? atoi(getenv("LLAMAHIP_B9S_MIN")) : (int)B9S_MIN_ROWS;   // (A/B switch of round 6's measurements: 65 = off)
    if (m.wtype == 0 && (int)n >= b9s_min && n <= B9S_MAX_ROWS) {   // fp32, 33..64 rows: the same schedule over planes on k_stream_b9 (nine exact bf16 products)
        const int rs = eval_q8b_layers(p, x, x_out_dev, n, past, last_row_only, bc);
        if (rs != ST_NA) return rs;   // (ST_NA comes back before anything is enqueued: the fp32-MFMA route below takes the Eval)
    }
    // fp32 prompts just past the stream kernels' 128 rows: two passes over the weights, ceil(n / 2) rows each (the rows of a pass are independent of the other
    // pass's in every launch but attention, which runs once over all n as before).  Measurements: STREAM_TWO_PASS_MAX.
    static const uint32_t two_pass_max = getenv("LLAMAHIP_TWO_PASS_MAX") ? (uint32_t)atoi(getenv("LLAMAHIP_TWO_PASS_MAX")) : STREAM_TWO_PASS_MAX;   // (A/B switch: 128 = off)
    const bool two_pass = !bc && m.wtype == 0 && n > stream_max_rows() && n <= two_pass_max && n <= 2 * stream_max_rows() && d % GBK == 0 && F % GBK == 0 && stream_shape_ok(ctx, m);
    const uint32_t sb = two_pass ? (n + 1) / 2 : n;   // rows per pass of the stream kernels
    bool h_ready = false;   // p->h already holds this layer's RMSNorm * attn_norm rows (written by the previous layer's w2 reduce pass)
    for (uint32_t il = m.layer0; il < m.layer1; ++il) {
        const LayerW& L = m.layers[il];
        const size_t slot = (size_t)(il - m.cache_layer0) * m.ctx * d;
        const bool mfma = (n >= MFMA_MIN_ROWS || (n >= 2 && m.wtype == 0)) && d % GBK == 0 && F % GBK == 0;   // grouped MFMA launches: from 9 rows (tile GEMM), from 2 rows on the streaming MFMA kernel
        const bool q8 = m.wtype == 7;
        bool qkv_roped = false, gated = false;
        // 17..64 rows: wo / w2 as K-split pairs whose reduce pass also writes the next norm's rows into p->h (gemm_stream_split)
        const bool ksp = sb > 16 && sb <= stream_max_rows() && !q8 && mfma;
        bool hf_ready = false;
        const float* wqkv[3] = {L.wq, L.wk, L.wv};
        const float* sqkv[3] = {L.s_wq, L.s_wk, L.s_wv};
        float* yqkv[3] = {p->qraw, p->kraw, p->vraw};
        // RoPE + cache append in the GEMM's epilogue (no rope_store pass, no raw q/k/v round trip): where each row goes
        StreamArgs fq = {};
        fq.epi = ST_EPI_QKV_ROPE; fq.q_out = p->q; fq.k_cache = m.kc + slot; fq.v_cache = m.vc + slot; fq.rope = rope; fq.hd = m.hd; fq.past = past;
        fq.rows = rows; fq.kv_off = slot;
        const bool fold_norm = n <= 16 && !q8 && mfma;
        if (n <= stream_max_rows() && fold_norm) {
            // short prompts: ONE launch for RMSNorm (folded: gamma at staging, the per-token scale in the epilogue) -> wq|wk|wv -> RoPE -> cache append
            // (folded up to 16 rows: -3..5 % per Eval; at 17..32 rows the extra staging work of the loader waves eats the saved launch)
            StreamArgs fa = fq;
            fa.gamma = L.attn_norm;
            const int rs = gemm_stream_group(ctx, x, d, 3, wqkv, nullptr, nullptr, d, d, n, d, "stream_norm_wqkv_rope", &fa);
            if (rs < 0) return rs;
            qkv_roped = rs == 0;
        }
        if (!qkv_roped && !h_ready) { TraceScope ts_(ctx->stream, "rmsnorm_rows_a"); LH_LAUNCH(k_rmsnorm_rows, dim3(n), dim3(256), 0, ctx->stream, x, L.attn_norm, p->h, d); }
        h_ready = false;
        if (qkv_roped) {
        } else if (q8) {   // (prompts beyond k_stream_q8b's 64 rows)
            if ((rc = gemm_q8_group(ctx, p->h, d, 3, wqkv, sqkv, yqkv, nullptr, d, d, n, d, "gemm_q8_wqkv"))) return rc;
        } else if (mfma) {
            int rs = ST_NA;
            for (uint32_t b0 = 0; b0 < n && sb <= stream_max_rows(); b0 += sb) {
                StreamArgs fb = fq;
                fb.q_out = p->q + (size_t)b0 * d; fb.past = past + b0;
                if ((rs = gemm_stream_group(ctx, p->h + (size_t)b0 * d, d, 3, wqkv, nullptr, nullptr, d, d, std::min(sb, n - b0), d, "stream_wqkv_rope", &fb))) break;
            }
            if (rs == ST_NA && n > 64) {   // long prompts: the same epilogue in the tile GEMM
                GemmArgs ga = {};
                ga.epi = GEMM_EPI_QKV_ROPE; ga.q_out = p->q; ga.k_cache = m.kc + slot; ga.v_cache = m.vc + slot; ga.rope = rope; ga.hd = m.hd; ga.past = past;
                rs = gemm_mfma_group(ctx, p->h, d, 3, wqkv, nullptr, nullptr, d, d, n, d, "gemm_wqkv_rope", &ga);
            }
            if (rs < 0) return rs;
            qkv_roped = rs == 0;
            if (!qkv_roped && (rc = gemm_mfma_group(ctx, p->h, d, 3, wqkv, yqkv, nullptr, d, d, n, d, "gemm_wqkv"))) return rc;
        } else {
            if ((rc = gemm_small_n(ctx, L.wq, p->h, p->qraw, nullptr, d, d, n, d, d, "gemm_wq"))) return rc;
            if ((rc = gemm_small_n(ctx, L.wk, p->h, p->kraw, nullptr, d, d, n, d, d, "gemm_wk"))) return rc;
            if ((rc = gemm_small_n(ctx, L.wv, p->h, p->vraw, nullptr, d, d, n, d, d, "gemm_wv"))) return rc;
        }
        if (!qkv_roped) { TraceScope ts_(ctx->stream, "rope_store"); LH_LAUNCH(k_rope_store, dim3(n), dim3(256), 0, ctx->stream, (const float*)p->qraw, (const float*)p->kraw, (const float*)p->vraw, p->q, m.kc + slot,
                           m.vc + slot, rope, d, m.hd, past, rows, (uint64_t)slot); }
        if (rows) {   // rows of different streams: one query each, against its own cache up to its own position
            AttnArgs a = {};
            a.q = p->q; a.out = p->attn; a.d = d; a.hd = m.hd; a.n = n; a.scale = scale; a.rows = rows; a.kv_off = slot;
            if (bc->attn_part) { if ((rc = launch_attention_split(p, a, bc->attn_part))) return rc; }
            else if ((rc = launch_attention(ctx, a, m.ctx))) return rc;
        } else if (mfma && n >= 32 && m.hd == FA_HD) {   // single pass, online softmax
            if ((rc = attention_flash(p, p->q, m.kc + slot, m.vc + slot, p->attn, n, past, scale))) return rc;
        } else if (mfma && n >= 32 && m.hd % 32 == 0) {  // other head sizes: batched MFMA GEMMs over heads with a score tensor
            if ((rc = attention_gemm(p, p->q, m.kc + slot, m.vc + slot, p->attn, n, past, scale))) return rc;
        } else {   // fewer queries: the per-query kernel (no score tensor) is cheaper
            AttnArgs a = {};
            a.q = p->q; a.k_cache = m.kc + slot; a.v_cache = m.vc + slot; a.out = p->attn; a.d = d; a.hd = m.hd; a.n = n; a.scale = scale; a.sp = nullptr; a.past_host = past;
            if ((rc = launch_attention(ctx, a, past + n))) return rc;
        }
        int wo_rs = ST_NA;
        if (ksp) {
            for (uint32_t b0 = 0; b0 < n; b0 += sb)
                if ((wo_rs = gemm_stream_split(ctx, L.wo, p->attn + (size_t)b0 * d, d, d, d, std::min(sb, n - b0), x + (size_t)b0 * d, p->xb + (size_t)b0 * d, L.ffn_norm, p->h + (size_t)b0 * d, "stream_wo_ksplit"))) break;
            if (wo_rs < 0) return wo_rs;
            hf_ready = wo_rs == 0;
        }
        if (wo_rs == 0) {
        } else if (q8) { if ((rc = gemm_q8(ctx, L.wo, L.s_wo, p->attn, p->xb, x, d, d, n, d, d, "gemm_q8_wo"))) return rc; }
        else if ((rc = gemm_small_n(ctx, L.wo, p->attn, p->xb, x, d, d, n, d, d, "gemm_wo"))) return rc;
        const float* w13[2] = {L.w1, L.w3};
        const float* s13[2] = {L.s_w1, L.s_w3};
        float* y13[2] = {p->a1, p->a3};
        float* yg[2] = {p->g, nullptr};
        if (n <= stream_max_rows() && fold_norm) {   // short prompts: RMSNorm (folded) -> w1|w3 -> silu * mul in one launch
            StreamArgs fa = {};
            fa.epi = ST_EPI_SILU_MUL;
            fa.gamma = L.ffn_norm;
            const int rs = gemm_stream_group(ctx, p->xb, d, 2, w13, yg, nullptr, F, d, n, F, "stream_norm_w1w3_silu", &fa);
            if (rs < 0) return rs;
            gated = rs == 0;
        }
        if (!gated && !hf_ready) { TraceScope ts_(ctx->stream, "rmsnorm_rows_f"); LH_LAUNCH(k_rmsnorm_rows, dim3(n), dim3(256), 0, ctx->stream, (const float*)p->xb, L.ffn_norm, p->h, d); }
        if (gated) {
        } else if (q8) {
            if ((rc = gemm_q8_group(ctx, p->h, d, 2, w13, s13, y13, nullptr, F, d, n, F, "gemm_q8_w1w3"))) return rc;
        } else if (mfma) {
            StreamArgs fa = {};   // short prompts: silu(w1 h) * (w3 h) in the epilogue of (w1, w3) tile pairs
            fa.epi = ST_EPI_SILU_MUL;
            int rs = ST_NA;
            for (uint32_t b0 = 0; b0 < n && sb <= stream_max_rows(); b0 += sb) {
                float* ygb[2] = {p->g + (size_t)b0 * F, nullptr};
                if ((rs = gemm_stream_group(ctx, p->h + (size_t)b0 * d, d, 2, w13, ygb, nullptr, F, d, std::min(sb, n - b0), F, "stream_w1w3_silu", &fa))) break;
            }
            if (rs == ST_NA && n > 64) {
                GemmArgs ga = {};
                ga.epi = GEMM_EPI_SILU_MUL;
                rs = gemm_mfma_group(ctx, p->h, d, 2, w13, yg, nullptr, F, d, n, F, "gemm_w1w3_silu", &ga);
            }
            if (rs < 0) return rs;
            gated = rs == 0;
            if (!gated && (rc = gemm_mfma_group(ctx, p->h, d, 2, w13, y13, nullptr, F, d, n, F, "gemm_w1w3"))) return rc;
        } else {
            if ((rc = gemm_small_n(ctx, L.w1, p->h, p->a1, nullptr, F, d, n, d, F, "gemm_w1"))) return rc;
            if ((rc = gemm_small_n(ctx, L.w3, p->h, p->a3, nullptr, F, d, n, d, F, "gemm_w3"))) return rc;
        }
        if (!gated) { TraceScope ts_(ctx->stream, "silu_mul"); LH_LAUNCH(k_silu_mul, dim3(std::min<uint64_t>(((uint64_t)n * F + 255) / 256, 4096)), dim3(256), 0, ctx->stream, (const float*)p->a1,
                           (const float*)p->a3, p->g, (uint64_t)n * F); }
        const bool last = il + 1 == m.layer1;
        float* y = (last && !m.last_stage()) ? x_out_dev : p->xa;
        int w2_rs = ST_NA;
        if (ksp) {
            const float* next_gamma = last ? nullptr : m.layers[il + 1].attn_norm;   // the next layer's first norm rides on the reduce pass
            for (uint32_t b0 = 0; b0 < n; b0 += sb)
                if ((w2_rs = gemm_stream_split(ctx, L.w2, p->g + (size_t)b0 * F, F, d, F, std::min(sb, n - b0), p->xb + (size_t)b0 * d, y + (size_t)b0 * d, next_gamma, p->h + (size_t)b0 * d, "stream_w2_ksplit"))) break;
            if (w2_rs < 0) return w2_rs;
            h_ready = w2_rs == 0 && next_gamma != nullptr;
        }
        if (w2_rs == 0) {
        } else if (q8) { if ((rc = gemm_q8(ctx, L.w2, L.s_w2, p->g, y, p->xb, d, F, n, F, d, "gemm_q8_w2"))) return rc; }
        else if ((rc = gemm_small_n(ctx, L.w2, p->g, y, p->xb, d, F, n, F, d, "gemm_w2"))) return rc;
        x = p->xa;
        LH_HIP(ctx, hipGetLastError());
    }
    if (m.last_stage()) {
        // the reference evaluates norm + lm_head for all N rows (llama.go:372-384) although only row N-1 is read (llama.go:394-401);
        // callers that say so (LH_GRAPH_LAST_ROW_LOGITS, the lh_llama_* entry points) get that row only, in its usual place
        const uint32_t r0 = last_row_only ? n - 1 : 0, nr = n - r0;
        { TraceScope ts_(ctx->stream, "rmsnorm_rows_final"); LH_LAUNCH(k_rmsnorm_rows, dim3(nr), dim3(256), 0, ctx->stream, x + (size_t)r0 * d, m.norm, p->h + (size_t)r0 * d, d); }
        if (m.wtype == 7) {
            if (nr >= Q8_GEMM_MIN_ROWS) {
                if ((rc = gemm_q8(ctx, m.output, m.s_output, p->h + (size_t)r0 * d, p->logits + (size_t)r0 * m.V, nullptr, m.V, d, nr, d, m.V, "gemm_q8_lmhead"))) return rc;
            } else {   // a few rows (the last row of a long prompt): the int8 GEMV row by row
                for (uint32_t i = 0; i < nr; ++i) {
                    GemvArgs ga = {};
                    ga.w[0] = m.output; ga.ws[0] = m.s_output; ga.M = m.V; ga.K = d; ga.x = p->h + (size_t)(r0 + i) * d; ga.y = p->logits + (size_t)(r0 + i) * m.V;
                    if ((rc = gemv<PRO_PLAIN, EPI_STORE, MAP_SINGLE>(ctx, ga, "gemv_lmhead_row", 7))) return rc;
                }
            }
        } else if (nr == 1) {   // the one row llama.Eval reads: the decode weight stream (76 us on 7B; the column kernel took 122)
            GemvArgs ga = {};
            ga.w[0] = m.output; ga.M = m.V; ga.K = d; ga.x = p->h + (size_t)r0 * d; ga.y = p->logits + (size_t)r0 * m.V;
            if ((rc = gemv<PRO_PLAIN, EPI_STORE, MAP_SINGLE>(ctx, ga, "gemv_lmhead_row", 0))) return rc;
        } else if ((rc = gemm_small_n(ctx, m.output, p->h + (size_t)r0 * d, p->logits + (size_t)r0 * m.V, nullptr, m.V, d, nr, d, m.V, "gemm_lmhead"))) return rc;
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}


// ---- lh_batch: the pods of a rank in ONE weight pass ---------------------------------------------------------------------------
// The reference's only parallelism is request-level: Engine() runs up to MaxPods Do() goroutines (pkg/server/server.go:84-106), each with
// its own llama.Context over the shared Model (server.go:151).  On the GPU an N = 1 Eval streams all weights, so P pods that decode
// independently read P x 26.4 GB per round of tokens for the bandwidth of one.  A batch binds the stages of P streams (same weights and
// layer range, one KV cache each) and evaluates one decode step of ALL of them as a P-row pass: the stream GEMMs of the short-prompt path
// (k_stream_mm2: every weight byte read once for all rows), RoPE / cache append / attention per row from a device table
// {row -> its cache, its position}.  Row results do not depend on the row's index or on the other rows (every output column of the MFMA
// tiles accumulates on its own), so a stream decodes the same ids whoever shares its tick.
// One tick = one captured hipGraph (token ids, positions and the residual stream live at fixed device addresses; the graph itself moves the
// positions on), so a tick costs the host one hipGraphLaunch whatever the stage's length.
struct Batch {
    lh_ctx* ctx = nullptr;
    std::vector<Plan*> pods;
    uint32_t B = 0;
    bool batched = false;          // rows share one weight pass (else: row by row on their own plans - same results, P weight passes)
    BatchRow* rows_dev = nullptr;  // [B]
    uint32_t* tok_dev = nullptr;   // [B] token ids of the next tick (first stage)
    uint32_t* ids_dev = nullptr;   // [B] ids produced by the last tick (last stage)
    uint32_t* out_dev = nullptr;   // [B][out_cap] ids produced per row (last stage)
    uint32_t out_cap = 0;
    StepParams* sp_dev = nullptr;  // [B] the rows' step parameters, in lockstep with rows_dev (what the N = 1 kernels and the sampler read)
    float* logits_own = nullptr;   // [B][V] row-by-row mode with B > 1 (last stage)
    float* attn_part = nullptr;    // batched, ctx > 256: split-T attention partials for B rows
    // sampling ticks (lh_batch_set_sampler): per-row sampler state + lastNTokens ring
    SampleState* ss_dev = nullptr; // [B]
    uint32_t* ring_dev = nullptr;  // [B][ring_cap]
    uint32_t ring_cap = 0, smp_topk = 0;
    bool sampling = false;
    // the captured tick
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    const float* cap_x_in = nullptr;
    float* cap_x_out = nullptr;
    uint64_t cap_splitk_gen = 0, cap_scratch_gen = 0;
    uint64_t scratch_gen() const { uint64_t g = 0; for (const Plan* p : pods) g += p->scratch_gen; return g; }   // grows when any pod's scratch moved
    bool cap_sampling = false;
    bool warm = false;
    // host mirror of the rows' positions (rows_dev[i].pos): the tick kernels advance them in device memory with no bound of their own, so
    // the window check of every other Eval entry point (plan_eval: past + n <= ctx) is made here before a tick is enqueued
    std::vector<uint32_t> pos;
    bool pos_known = false;        // false until lh_batch_set / lh_batch_prompt placed the rows
    // Context swap (server.go:160-172) of whole-model batches.  The ids live on the device (out_dev: row i's list, entry step0 + t written by
    // tick t since the last batch_set); what the host needs at a swap it reads then: the token tick t evaluated is entry step0 + t - 1 (or
    // tok0[i], the host value of the last batch_set, for entry -1).
    std::vector<uint32_t> pos_set, tok0, pending;
    bool tok0_known = false;
    uint32_t step0 = 0, ticks = 0, drained = 0;   // ticks since the last batch_set / of them recorded in the pods' token histories
    bool collecting = false;                      // lh_batch_decode: keep every id a row produces across the resets of the device lists
    std::vector<std::vector<uint32_t>> gen;
    float* logits() const { return batched || B == 1 ? pods[0]->logits : logits_own; }
};

static void batch_drop_graph(Batch* b) {
    if (b->exec) { hipGraphExecDestroy(b->exec); b->exec = nullptr; }
    if (b->graph) { hipGraphDestroy(b->graph); b->graph = nullptr; }
}

// the kernels of one tick, in stream order
static int batch_enqueue_tick(Batch* b, const float* x_in, float* x_out) {
    lh_ctx* ctx = b->ctx;
    Plan* p0 = b->pods[0];
    const ModelDesc& m = p0->md;
    const uint32_t B = b->B;
    int rc;
    if (b->batched) {
        BatchCtx bc = {b->rows_dev, b->tok_dev, b->attn_part};
        if ((rc = plan_eval(p0, nullptr, x_in, x_out, B, 0, false, &bc))) return rc;
    } else {
        for (uint32_t i = 0; i < B; ++i) {
            Plan* p = b->pods[i];
            if ((rc = enqueue_decode(p, b->sp_dev + i, x_in ? x_in + (size_t)i * m.d : nullptr, x_out ? x_out + (size_t)i * m.d : nullptr, 0, nullptr, b->tok_dev + i))) return rc;
            if (m.last_stage() && B > 1) LH_HIP(ctx, hipMemcpyAsync(b->logits_own + (size_t)i * m.V, p->logits, (size_t)m.V * 4, hipMemcpyDeviceToDevice, ctx->stream));
        }
    }
    if (m.last_stage() && b->sampling) {
        // SampleTopPTopK per row (llama.go:455-707) on the row's own ring / draw counter; the rows' step parameters carry the output index
        for (uint32_t i = 0; i < B; ++i)
            if ((rc = sample_launch(ctx, b->logits() + (size_t)i * m.V, m.V, b->ss_dev + i, b->ring_dev + (size_t)i * b->ring_cap, b->sp_dev + i, b->out_dev + (size_t)i * b->out_cap,
                                    nullptr, nullptr, nullptr, nullptr, 1, b->smp_topk)))
                return rc;
        LH_LAUNCH(k_batch_from_sp, dim3(1), dim3(64), 0, ctx->stream, b->rows_dev, b->tok_dev, b->ids_dev, (const StepParams*)b->sp_dev, B);
    } else if (m.last_stage()) {
        LH_LAUNCH(k_batch_argmax, dim3(B), dim3(1024), 0, ctx->stream, (const float*)b->logits(), m.V, b->rows_dev, b->tok_dev, b->ids_dev, b->out_dev, b->out_cap,
                           b->sp_dev, 1);
    } else {
        LH_LAUNCH(k_batch_advance, dim3(1), dim3(64), 0, ctx->stream, b->rows_dev, b->sp_dev, B);
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

// One tick on the stream: the first one eagerly (it sets kernel attributes and makes the allocations a capture must not make), then a
// captured graph, re-captured when an address it holds has changed.
static int batch_tick_unchecked(Batch* b, const float* x_in, float* x_out);
static int batch_swap(Batch* b);
static int batch_tick(Batch* b, const float* x_in, float* x_out) {
    lh_ctx* ctx = b->ctx;
    const ModelDesc& m = b->pods[0]->md;
    // a tick evaluates row i at position pos[i]: RoPE table row, KV append and the attention's key range all index by it (Eval's
    // pastCount + N <= CtxSize, checked like plan_eval / lh_llama_stage do)
    if (!b->pos_known) LH_FAIL(ctx, LH_EINVAL, "lh_batch: tick before lh_batch_set / lh_batch_prompt placed the rows");
    bool full = false;
    for (uint32_t i = 0; i < b->B; ++i) full = full || b->pos[i] >= m.ctx;
    int rc;
    if (full && (rc = batch_swap(b))) return rc;   // whole-model batches swap context like server.Do; a stage's batch fails here
    rc = batch_tick_unchecked(b, x_in, x_out);
    if (rc == 0) { for (uint32_t i = 0; i < b->B; ++i) b->pos[i] += 1; b->ticks += 1; }
    return rc;
}
static int batch_tick_unchecked(Batch* b, const float* x_in, float* x_out) {
    lh_ctx* ctx = b->ctx;
    Plan* p0 = b->pods[0];
    int rc;
    if (!p0->use_graph || !b->warm) {
        b->warm = true;
        return batch_enqueue_tick(b, x_in, x_out);
    }
    if (b->exec && (b->cap_x_in != x_in || b->cap_x_out != x_out || b->cap_splitk_gen != ctx->splitk_gen || b->cap_scratch_gen != b->scratch_gen() || b->cap_sampling != b->sampling))
        batch_drop_graph(b);
    if (!b->exec) {
        LH_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
        rc = batch_enqueue_tick(b, x_in, x_out);
        hipError_t e = hipStreamEndCapture(ctx->stream, &b->graph);
        if (rc) { if (b->graph) { hipGraphDestroy(b->graph); b->graph = nullptr; } return rc; }
        if (e != hipSuccess) LH_FAIL(ctx, LH_EHIP, "hipStreamEndCapture (batch tick): %s", hipGetErrorString(e));
        LH_HIP(ctx, hipGraphInstantiate(&b->exec, b->graph, nullptr, nullptr, 0));
        b->cap_x_in = x_in; b->cap_x_out = x_out; b->cap_splitk_gen = ctx->splitk_gen; b->cap_scratch_gen = b->scratch_gen(); b->cap_sampling = b->sampling;
    }
    LH_HIP(ctx, hipGraphLaunch(b->exec, ctx->stream));
    return 0;
}

static void batch_free(Batch* b) {
    if (!b) return;
    hipSetDevice(b->ctx->device);
    hipStreamSynchronize(b->ctx->stream);
    batch_drop_graph(b);
    void* bufs[] = {b->rows_dev, b->tok_dev, b->ids_dev, b->out_dev, b->sp_dev, b->logits_own, b->attn_part, b->ss_dev, b->ring_dev};
    for (void* q : bufs) if (q) hipFree(q);
    delete b;
}

// positions (and, tokens != nullptr, token ids) of the next tick from host values; the tick counter restarts at step0
static int batch_set(Batch* b, const uint32_t* tokens, const uint32_t* past, uint32_t step0 = 0) {
    lh_ctx* ctx = b->ctx;
    const ModelDesc& m = b->pods[0]->md;
    BatchSetArgs v = {};
    for (uint32_t i = 0; i < b->B; ++i) {
        // past == ctx is a legal resting place (a prompt that filled the window, Eval's pastCount + N <= CtxSize): a tick from there is refused by batch_tick
        if (past[i] > m.ctx) LH_FAIL(ctx, LH_EINVAL, "lh_batch_set: row %u at position %u outside the context window of %u", i, past[i], m.ctx);
        if (tokens && tokens[i] >= m.V) LH_FAIL(ctx, LH_EINVAL, "lh_batch_set: token id %u of row %u outside the vocabulary of %u", tokens[i], i, m.V);
        v.pos[i] = past[i];
        v.tok[i] = tokens ? tokens[i] : 0;
    }
    LH_LAUNCH(k_batch_set, dim3(1), dim3(64), 0, ctx->stream, b->rows_dev, b->tok_dev, b->B, tokens ? 1 : 0, v, b->sp_dev, step0);
    LH_HIP(ctx, hipGetLastError());
    b->pos.assign(past, past + b->B);
    b->pos_known = true;
    b->pos_set = b->pos;
    b->tok0_known = tokens != nullptr;
    if (tokens) b->tok0.assign(tokens, tokens + b->B);
    b->step0 = step0; b->ticks = 0; b->drained = 0;
    return 0;
}

// Waits for the stream and brings the host up to date with what the ticks since the last batch_set did: the token every tick evaluated goes
// into its pod's history (Plan::hist), the ids produced into b->gen (lh_batch_decode), the rows' pending tokens into b->pending.
static int batch_drain(Batch* b) {
    lh_ctx* ctx = b->ctx;
    const uint32_t B = b->B, n_out = b->step0 + b->ticks;
    const ModelDesc& m = b->pods[0]->md;
    if (!m.last_stage()) LH_FAIL(ctx, LH_EINVAL, "lh_batch: only the last stage knows the ids");
    std::vector<uint32_t> out((size_t)B * std::max(n_out, 1u));
    if (n_out) LH_HIP(ctx, hipMemcpy2DAsync(out.data(), (size_t)n_out * 4, b->out_dev, (size_t)b->out_cap * 4, (size_t)n_out * 4, B, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    b->pending.assign(B, Plan::HIST_UNKNOWN);
    for (uint32_t i = 0; i < B; ++i) {
        auto evaluated = [&](uint32_t t) -> uint32_t {   // the token tick t evaluated
            if (b->step0 + t >= 1) return out[(size_t)i * n_out + b->step0 + t - 1];
            return b->tok0_known ? b->tok0[i] : Plan::HIST_UNKNOWN;
        };
        for (uint32_t t = b->drained; t < b->ticks; ++t) {
            b->pods[i]->record(b->pos_set[i] + t, evaluated(t));
            if (b->collecting) b->gen[i].push_back(out[(size_t)i * n_out + b->step0 + t]);
        }
        b->pending[i] = evaluated(b->ticks);
    }
    b->drained = b->ticks;
    return 0;
}

// The rows whose window is full swap context (server.go:160-172), each on its own plan and cache like its prompt ran: the run of
// (position - keep) / 2 tokens is re-fed as ONE Eval at position keep, and the row's pending token then takes the next tick at the position
// behind it - together exactly the reference's swap Eval (the pending token last in it), cut in two at its last row.
static int batch_swap(Batch* b) {
    lh_ctx* ctx = b->ctx;
    const ModelDesc& m = b->pods[0]->md;
    if (!m.first_stage() || !m.last_stage())
        LH_FAIL(ctx, LH_EINVAL, "lh_batch: a row has left the context window of %u (the context swap of a layer shard is driven by the pipeline, not by a stage)", m.ctx);
    int rc;
    if ((rc = batch_drain(b))) return rc;
    std::vector<uint32_t> newpos = b->pos, refeed;
    for (uint32_t i = 0; i < b->B; ++i) {
        if (b->pos[i] < m.ctx) continue;
        Plan* p = b->pods[i];
        if (p->keep >= m.ctx) LH_FAIL(ctx, LH_EINVAL, "context swap: KeepCount %u leaves no room in a window of %u", p->keep, m.ctx);
        if (b->pending[i] == Plan::HIST_UNKNOWN || !swap_refeed_tokens(p, b->pos[i], b->pending[i], &refeed))
            LH_FAIL(ctx, LH_EINVAL, "lh_batch: row %u has left the context window of %u and the tokens of its window are not known to the batch (run its prompt through lh_batch_prompt)", i, m.ctx);
        if (!refeed.empty() && (rc = plan_eval(p, refeed.data(), nullptr, nullptr, (uint32_t)refeed.size(), p->keep, true))) return rc;
        newpos[i] = p->keep + (uint32_t)refeed.size();
    }
    const std::vector<uint32_t> toks = b->pending;   // (batch_set overwrites b->tok0 from it)
    return batch_set(b, toks.data(), newpos.data(), 0);
}

}  // namespace lh

using namespace lh;

// ======================================================================================================
// C-ABI
// ======================================================================================================
extern "C" {

void lh_ctx_destroy(lh_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    destroy_plans(ctx);
    if (ctx->arena) hipFree(ctx->arena);
    if (ctx->splitk) hipFree(ctx->splitk);
    if (ctx->xs3) hipFree(ctx->xs3);
    if (ctx->tc_ev0) { hipEventDestroy(ctx->tc_ev0); hipEventDestroy(ctx->tc_ev1); }
    if (ctx->staging) hipHostFree(ctx->staging);
    if (ctx->out_pinned) hipHostFree(ctx->out_pinned);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

int lh_llama_create(lh_ctx* ctx, const lh_llama_desc* desc, lh_llama** out) {
    if (!ctx || !desc || !out) LH_FAIL(ctx, LH_EINVAL, "lh_llama_create: NULL argument");
    *out = nullptr;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (desc->weight_dtype != 0 && desc->weight_dtype != 7) LH_FAIL(ctx, LH_EUNSUPPORTED, "lh_llama_create: weight dtype %d not supported", desc->weight_dtype);
    ModelDesc md;
    md.V = desc->vocab; md.d = desc->embd; md.H = desc->heads; md.L = desc->layers; md.F = desc->ff; md.ctx = desc->ctx;
    if (!md.H || !md.d || md.d % md.H) LH_FAIL(ctx, LH_ESHAPE, "lh_llama_create: bad embd/heads");
    md.hd = md.d / md.H;
    md.layer0 = desc->layer0; md.layer1 = desc->layer1 ? desc->layer1 : desc->layers;
    if (md.layer0 >= md.layer1 || md.layer1 > md.L) LH_FAIL(ctx, LH_EINVAL, "lh_llama_create: bad layer range [%u,%u)", md.layer0, md.layer1);
    md.cache_layer0 = md.layer0;
    md.wtype = desc->weight_dtype;
    auto need = [&](lh_buf b, uint64_t n, const char* what, const float** dst) -> int {
        Buffer* bf = find_buffer(ctx->ds, b);
        if (!bf) LH_FAIL(ctx, LH_EINVAL, "lh_llama_create: %s is not a registered buffer", what);
        if (bf->nfloats < n) LH_FAIL(ctx, LH_ESHAPE, "lh_llama_create: %s has %llu floats, needs %llu", what, (unsigned long long)bf->nfloats, (unsigned long long)n);
        if (bf->dtype != 0) LH_FAIL(ctx, LH_ESHAPE, "lh_llama_create: %s must be f32", what);
        *dst = bf->dev;
        return 0;
    };
    // weight matrix: f32, or block-int8 planes when the model is quantised
    auto needw = [&](lh_buf b, uint64_t rows, uint64_t cols, const char* what, const float** dst, const float** sc) -> int {
        Buffer* bf = find_buffer(ctx->ds, b);
        if (!bf) LH_FAIL(ctx, LH_EINVAL, "lh_llama_create: %s is not a registered buffer", what);
        if (bf->nfloats < rows * cols || bf->dtype != md.wtype) LH_FAIL(ctx, LH_ESHAPE, "lh_llama_create: %s has the wrong size or dtype", what);
        if (md.wtype == 7 && (bf->rows != rows || bf->cols != cols)) LH_FAIL(ctx, LH_ESHAPE, "lh_llama_create: %s block-int8 shape mismatch", what);
        *dst = bf->dev;
        *sc = bf->scales;
        return 0;
    };
    int rc;
    const uint64_t d = md.d, F = md.F, V = md.V;
    if (md.first_stage() && (rc = need(desc->tok_embeddings, V * d, "tok_embeddings", &md.tok_emb))) return rc;
    if (md.last_stage()) {
        if ((rc = need(desc->norm, d, "norm", &md.norm))) return rc;
        if ((rc = needw(desc->output, V, d, "output", &md.output, &md.s_output))) return rc;
    }
    md.layers.resize(md.L);
    for (uint32_t il = md.layer0; il < md.layer1; ++il) {
        const lh_llama_layer& s = desc->layer[il];
        LayerW& L = md.layers[il];
        if ((rc = need(s.attention_norm, d, "attention_norm", &L.attn_norm))) return rc;
        if ((rc = needw(s.wq, d, d, "wq", &L.wq, &L.s_wq))) return rc;
        if ((rc = needw(s.wk, d, d, "wk", &L.wk, &L.s_wk))) return rc;
        if ((rc = needw(s.wv, d, d, "wv", &L.wv, &L.s_wv))) return rc;
        if ((rc = needw(s.wo, d, d, "wo", &L.wo, &L.s_wo))) return rc;
        if ((rc = need(s.ffn_norm, d, "ffn_norm", &L.ffn_norm))) return rc;
        if ((rc = needw(s.w1, F, d, "w1", &L.w1, &L.s_w1))) return rc;
        if ((rc = needw(s.w2, d, F, "w2", &L.w2, &L.s_w2))) return rc;
        if ((rc = needw(s.w3, F, d, "w3", &L.w3, &L.s_w3))) return rc;
    }
    const float *kc, *vc;
    const uint64_t kvn = d * (md.layer1 - md.layer0) * md.ctx;
    if ((rc = need(desc->k_cache, kvn, "k_cache", &kc))) return rc;
    if ((rc = need(desc->v_cache, kvn, "v_cache", &vc))) return rc;
    md.kc = (float*)kc; md.vc = (float*)vc;
    {   // the cache's token history travels with the buffer: shared with the graph path's plan over the same cache
        Buffer* kb = find_buffer(ctx->ds, desc->k_cache);
        if (!kb->kv_hist) kb->kv_hist = std::make_shared<std::vector<uint32_t>>();
        md.kv_hist = kb->kv_hist;
    }
    Plan* p = nullptr;
    if ((rc = plan_create(ctx, md, &p))) return rc;
    lh_llama* m = new lh_llama();
    m->ctx = ctx;
    m->plan = p;
    *out = m;
    return LH_OK;
}

void lh_llama_destroy(lh_llama* m) {
    if (!m) return;
    plan_destroy(m->plan);
    delete m;
}

int lh_llama_eval(lh_llama* m, const uint32_t* tokens, uint32_t n, uint32_t past, float* logits_host) {
    if (!m || !tokens) return LH_EINVAL;
    lh_ctx* ctx = m->ctx;
    Plan* p = m->plan;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (!p->md.first_stage() || !p->md.last_stage()) LH_FAIL(ctx, LH_EINVAL, "lh_llama_eval needs a whole-model plan; use lh_llama_stage");
    int rc = plan_eval(p, tokens, nullptr, nullptr, n, past, true);
    if (rc) return rc;
    if (logits_host)
        LH_HIP(ctx, hipMemcpyAsync(logits_host, p->logits + (size_t)(n - 1) * p->md.V, (size_t)p->md.V * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_llama_set_keep(lh_llama* m, uint32_t keep) {
    if (!m) return LH_EINVAL;
    m->plan->keep = keep;
    return LH_OK;
}

int lh_llama_decode_greedy(lh_llama* m, uint32_t first_token, uint32_t past, uint32_t n_steps, uint32_t* out_tokens, float* logits_last_host) {
    if (!m || !n_steps) return LH_EINVAL;
    lh_ctx* ctx = m->ctx;
    Plan* p = m->plan;
    const ModelDesc& md = p->md;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (!md.first_stage() || !md.last_stage()) LH_FAIL(ctx, LH_EINVAL, "lh_llama_decode_greedy needs a whole-model plan");
    if (past > md.ctx) LH_FAIL(ctx, LH_EINVAL, "decode: position %u outside the context window of %u", past, md.ctx);
    if (first_token >= md.V) LH_FAIL(ctx, LH_EINVAL, "decode: token id %u outside the vocabulary of %u", first_token, md.V);
    int rc;
    if ((rc = ensure_out_tokens(p, n_steps))) return rc;
    if ((rc = upload_step_params(p, 0, first_token, past, 0))) return rc;
    // past + n_steps > ctx: the loop swaps context like server.Do does (server.go:160-172) instead of failing
    if ((rc = resident_steps_swapping(p, first_token, &past, 0, n_steps, false))) return rc;
    if (out_tokens) LH_HIP(ctx, hipMemcpyAsync(out_tokens, p->out_tokens_dev, (size_t)n_steps * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (logits_last_host) LH_HIP(ctx, hipMemcpyAsync(logits_last_host, p->logits, (size_t)md.V * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_llama_decode_sample(lh_llama* m, const uint32_t* prompt, uint32_t n_prompt, uint32_t n_predict, uint32_t ring_size, const lh_sample_params* sp,
                           uint32_t* out_tokens) {
    if (!m) return LH_EINVAL;
    lh_ctx* ctx = m->ctx;
    Plan* p = m->plan;
    const ModelDesc& md = p->md;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (!prompt || !n_prompt || !n_predict || !out_tokens) LH_FAIL(ctx, LH_EINVAL, "lh_llama_decode_sample: empty prompt, no tokens to predict or null output");
    if (!md.first_stage() || !md.last_stage()) LH_FAIL(ctx, LH_EINVAL, "lh_llama_decode_sample needs a whole-model plan");
    if (ring_size == 0) LH_FAIL(ctx, LH_EINVAL, "lh_llama_decode_sample: the lastNTokens ring needs at least one slot (the reference uses CtxSize, server.go:127)");
    if (n_prompt > md.ctx) LH_FAIL(ctx, LH_EINVAL, "decode: a prompt of %u tokens exceeds the context window of %u", n_prompt, md.ctx);
    int rc;
    if ((rc = sample_check(ctx, sp, md.V))) return rc;
    if ((rc = ensure_out_tokens(p, n_predict))) return rc;
    if (!p->ss_dev) LH_HIP(ctx, hipMalloc((void**)&p->ss_dev, sizeof(SampleState)));
    if ((sp->top_k <= 64) != (p->smp_topk <= 64)) drop_graphs(p, GM_SMP);  // the captured graphs hold the kernel variant
    p->smp_topk = sp->top_k;
    if (ring_size > p->ring_cap) {  // the captured sampler holds the ring pointer
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (p->ring_dev) LH_HIP(ctx, hipFree(p->ring_dev));
        p->ring_dev = nullptr;
        p->ring_cap = 0;
        LH_HIP(ctx, hipMalloc((void**)&p->ring_dev, (size_t)ring_size * 4));
        p->ring_cap = ring_size;
        drop_graphs(p, GM_SMP);
    }
    // ring: ring_size zeros, then the prompt ids (appendToken, server.go:129-138, 193-197): what remains is the last ring_size of them
    {
        std::vector<uint32_t> ring(ring_size, 0u);
        for (uint32_t i = 0; i < n_prompt; ++i) ring[i % ring_size] = prompt[i];
        SampleState st = {sp->top_k, sp->top_p, sp->temp, sp->repeat_penalty, sp->seed, 0, ring_size, n_prompt};
        if ((rc = ensure_staging(ctx, (uint64_t)ring_size * 4 + sizeof st))) return rc;
        LH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // staging reuse
        memcpy(ctx->staging, ring.data(), (size_t)ring_size * 4);
        memcpy((char*)ctx->staging + (size_t)ring_size * 4, &st, sizeof st);
        LH_HIP(ctx, hipMemcpyAsync(p->ring_dev, ctx->staging, (size_t)ring_size * 4, hipMemcpyHostToDevice, ctx->stream));
        LH_HIP(ctx, hipMemcpyAsync(p->ss_dev, (char*)ctx->staging + (size_t)ring_size * 4, sizeof st, hipMemcpyHostToDevice, ctx->stream));
    }
    if ((rc = plan_eval(p, prompt, nullptr, nullptr, n_prompt, 0, true))) return rc;
    // first sample on the last prompt row; the bookkeeping moves {past: n_prompt - 1, step: 0} to {token, past: n_prompt, step: 1}
    if ((rc = upload_step_params(p, 0, 0, n_prompt - 1, 0))) return rc;
    const float* last_row = n_prompt == 1 ? p->logits : p->logits + (size_t)(n_prompt - 1) * md.V;
    if ((rc = sample_launch(ctx, last_row, md.V, p->ss_dev, p->ring_dev, p->sp_dev, p->out_tokens_dev, nullptr, nullptr, nullptr, nullptr, 1, p->smp_topk))) return rc;
    if (n_predict > 1) {
        // the first id (entry 0 of the output list) is the first token the resident steps evaluate; past the window the loop swaps context
        // (server.go:160-172)
        uint32_t past = n_prompt;
        if ((rc = resident_steps_swapping(p, 0, &past, 1, n_predict - 1, true, /*first_in_out=*/true))) return rc;
    }
    LH_HIP(ctx, hipMemcpyAsync(out_tokens, p->out_tokens_dev, (size_t)n_predict * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_llama_stage(lh_llama* m, const uint32_t* tokens, const uint32_t* tokens_dev, const float* x_in_dev, float* x_out_dev, uint32_t n, uint32_t past,
                   float* logits_dev, uint32_t* argmax_dev) {
    if (!m) return LH_EINVAL;
    lh_ctx* ctx = m->ctx;
    Plan* p = m->plan;
    const ModelDesc& md = p->md;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if (n == 1) {
        if ((rc = plan_ensure_rows(p, 1))) return rc;
        if ((uint64_t)past + 1 > md.ctx) LH_FAIL(ctx, LH_EINVAL, "stage: position %u outside the context window", past);
        const uint32_t slot = 1 + (p->slot_counter++ % (SP_SLOTS - 1));
        if (md.first_stage() && !tokens && !tokens_dev) LH_FAIL(ctx, LH_EINVAL, "stage: first stage needs a token id (host or device)");
        if (md.first_stage() && tokens && tokens[0] >= md.V) LH_FAIL(ctx, LH_EINVAL, "stage: token id %u outside the vocabulary of %u", tokens[0], md.V);
        if ((rc = upload_step_params(p, slot, tokens ? tokens[0] : 0, past, 0))) return rc;
        if ((rc = enqueue_decode(p, p->sp_dev + slot, x_in_dev, x_out_dev, false, md.last_stage() ? argmax_dev : nullptr, tokens ? nullptr : tokens_dev))) return rc;
    } else {
        if (md.first_stage() && !tokens) LH_FAIL(ctx, LH_EINVAL, "stage: multi-row first stage needs host token ids");
        if ((rc = plan_eval(p, tokens, x_in_dev, x_out_dev, n, past, true))) return rc;
        if (md.last_stage() && argmax_dev) {
            LH_LAUNCH(k_argmax_advance, dim3(1), dim3(1024), 0, ctx->stream, (const float*)(p->logits + (size_t)(n - 1) * md.V), md.V, (StepParams*)nullptr,
                               (uint32_t*)nullptr, argmax_dev, 0);
            LH_HIP(ctx, hipGetLastError());
        }
    }
    if (md.last_stage() && logits_dev)
        LH_HIP(ctx, hipMemcpyAsync(logits_dev, p->logits + (size_t)(n - 1) * md.V, (size_t)md.V * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return LH_OK;
}

// ---- lh_batch ---------------------------------------------------------------------------------------------------------------------
struct lh_batch { lh::Batch* b; };

int lh_batch_create(lh_ctx* ctx, lh_llama* const* pods, uint32_t n_pods, lh_batch** out) {
    if (!ctx || !pods || !out) LH_FAIL(ctx, LH_EINVAL, "lh_batch_create: NULL argument");
    *out = nullptr;
    if (n_pods == 0 || n_pods > 64) LH_FAIL(ctx, LH_EINVAL, "lh_batch_create: %u rows outside 1..64", n_pods);
    LH_HIP(ctx, hipSetDevice(ctx->device));
    Batch* b = new Batch();
    b->ctx = ctx;
    b->B = n_pods;
    for (uint32_t i = 0; i < n_pods; ++i) {
        if (!pods[i] || pods[i]->ctx != ctx) { delete b; LH_FAIL(ctx, LH_EINVAL, "lh_batch_create: pod %u lives on another context (one stream orders the tick)", i); }
        Plan* p = pods[i]->plan;
        if (i > 0) {
            // same weights, same layer range, same shape; only the KV cache differs (a llama.Context per pod over one Model, server.go:151)
            ModelDesc a = p->md, c = b->pods[0]->md;
            a.kc = c.kc; a.vc = c.vc;
            if (!a.same(c)) { delete b; LH_FAIL(ctx, LH_EINVAL, "lh_batch_create: pod %u is not a stage of the same model as pod 0", i); }
            for (uint32_t j = 0; j < i; ++j)
                if (b->pods[j]->md.kc == p->md.kc || b->pods[j]->md.vc == p->md.vc) { delete b; LH_FAIL(ctx, LH_EINVAL, "lh_batch_create: pods %u and %u share a KV cache", j, i); }
        }
        b->pods.push_back(p);
    }
    Plan* p0 = b->pods[0];
    const ModelDesc& m = p0->md;
    b->batched = plan_batch_rows_ok(p0, n_pods);
    int rc = 0;
    if (b->batched) rc = plan_ensure_rows(p0, n_pods);
    else for (uint32_t i = 0; i < n_pods && !rc; ++i) rc = plan_ensure_rows(b->pods[i], 1);
    if (rc) { batch_free(b); return rc; }
    b->out_cap = m.ctx + 1;
    auto al = [&](void** ptr, size_t bytes) { return hipMalloc(ptr, bytes) == hipSuccess && hipMemsetAsync(*ptr, 0, bytes, ctx->stream) == hipSuccess; };
    bool ok = al((void**)&b->rows_dev, sizeof(BatchRow) * n_pods) && al((void**)&b->tok_dev, 4 * (size_t)n_pods) && al((void**)&b->ids_dev, 4 * (size_t)n_pods) &&
              al((void**)&b->sp_dev, sizeof(StepParams) * n_pods);
    if (ok && m.last_stage()) ok = al((void**)&b->out_dev, 4 * (size_t)n_pods * b->out_cap);
    if (ok && m.last_stage() && !b->batched && n_pods > 1) ok = al((void**)&b->logits_own, 4 * (size_t)n_pods * m.V);
    if (ok && b->batched && p0->attn_part) ok = al((void**)&b->attn_part, 4 * (size_t)n_pods * m.H * ((m.ctx + ATT_TC - 1) / ATT_TC) * (m.hd + 2));
    if (ok) {
        // The zero fills above run on the context's stream, which is non-blocking: a synchronous copy (null stream) is NOT ordered behind
        // them.  Without this wait the fill of rows_dev could land after the table below and leave null cache pointers in it (seen as a GPU
        // memory fault at cache position 8 x embd, twice in ~40 two-rank runs of round 3).
        ok = hipStreamSynchronize(ctx->stream) == hipSuccess;
        std::vector<BatchRow> hr(n_pods);
        for (uint32_t i = 0; i < n_pods; ++i) hr[i] = BatchRow{b->pods[i]->md.kc, b->pods[i]->md.vc, 0u, 0u};
        ok = ok && hipMemcpy(b->rows_dev, hr.data(), sizeof(BatchRow) * n_pods, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!ok || hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipGetLastError(); batch_free(b); LH_FAIL(ctx, LH_ENOMEM, "lh_batch_create: device allocation failed"); }
    lh_batch* h = new lh_batch();
    h->b = b;
    *out = h;
    return LH_OK;
}

void lh_batch_destroy(lh_batch* h) {
    if (!h) return;
    batch_free(h->b);
    delete h;
}

uint32_t lh_batch_rows(const lh_batch* h) { return h ? h->b->B : 0; }
int lh_batch_batched(const lh_batch* h) { return h && h->b->batched ? 1 : 0; }
uint32_t* lh_batch_tokens_dev(lh_batch* h) { return h ? h->b->tok_dev : nullptr; }
uint32_t* lh_batch_ids_dev(lh_batch* h) { return h ? h->b->ids_dev : nullptr; }

int lh_batch_read_ids(lh_batch* h, uint32_t* ids_host) {
    if (!h || !ids_host) return LH_EINVAL;
    lh_ctx* ctx = h->b->ctx;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    LH_HIP(ctx, hipMemcpyAsync(ids_host, h->b->ids_dev, (size_t)h->b->B * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_batch_set(lh_batch* h, const uint32_t* tokens, const uint32_t* past) {
    if (!h || !past) return LH_EINVAL;
    LH_HIP(h->b->ctx, hipSetDevice(h->b->ctx->device));
    return batch_set(h->b, tokens, past);
}

int lh_batch_set_sampler(lh_batch* h, const lh_sample_params* sp, uint32_t ring_size, const uint32_t* const* ring_init, const uint32_t* n_init) {
    if (!h) return LH_EINVAL;
    Batch* b = h->b;
    lh_ctx* ctx = b->ctx;
    const ModelDesc& m = b->pods[0]->md;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (!sp) { b->sampling = false; return LH_OK; }   // back to greedy ticks
    if (!m.last_stage()) { b->sampling = false; return LH_OK; }   // only the last stage samples; the other stages' ticks are the same either way
    int rc;
    if ((rc = sample_check(ctx, sp, m.V))) return rc;
    if (ring_size == 0) LH_FAIL(ctx, LH_EINVAL, "lh_batch_set_sampler: the lastNTokens ring needs at least one slot (the reference uses CtxSize, server.go:127)");
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!b->ss_dev) LH_HIP(ctx, hipMalloc((void**)&b->ss_dev, sizeof(SampleState) * b->B));
    if (ring_size > b->ring_cap) {
        if (b->ring_dev) LH_HIP(ctx, hipFree(b->ring_dev));
        b->ring_dev = nullptr; b->ring_cap = 0;
        LH_HIP(ctx, hipMalloc((void**)&b->ring_dev, (size_t)b->B * ring_size * 4));
        b->ring_cap = ring_size;
        batch_drop_graph(b);
    }
    if ((sp->top_k <= 64) != (b->smp_topk <= 64)) batch_drop_graph(b);   // the captured tick holds the kernel variant
    b->smp_topk = sp->top_k;
    // ring: ring_size zeros, then the ids seen so far (appendToken, server.go:129-138, 193-197)
    std::vector<uint32_t> ring((size_t)b->B * b->ring_cap, 0u);
    std::vector<SampleState> st(b->B);
    for (uint32_t i = 0; i < b->B; ++i) {
        const uint32_t ni = (ring_init && n_init) ? n_init[i] : 0;
        for (uint32_t j = 0; j < ni; ++j) ring[(size_t)i * b->ring_cap + j % ring_size] = ring_init[i][j];
        st[i] = SampleState{sp->top_k, sp->top_p, sp->temp, sp->repeat_penalty, sp->seed, 0, ring_size, ni};   // every row = a solo run with this seed
    }
    LH_HIP(ctx, hipMemcpy(b->ring_dev, ring.data(), ring.size() * 4, hipMemcpyHostToDevice));
    LH_HIP(ctx, hipMemcpy(b->ss_dev, st.data(), sizeof(SampleState) * b->B, hipMemcpyHostToDevice));
    // the rows' output lists restart (.step = index into them); token ids and positions stay what lh_batch_set / lh_batch_prompt / the ticks made
    // them.  The HOST mirror restarts with them: what the ticks so far produced is read first (the histories of the context swap and
    // each row's pending token), then the mirror is re-based on "tick 0 evaluates the pending token and writes entry 0" - the call may come
    // mid-stream, behind a prompt and any number of ticks.
    if (b->step0 + b->ticks > 0) {
        if ((rc = batch_drain(b))) return rc;
        b->pos_set = b->pos;
        b->tok0 = b->pending;
        b->tok0_known = true;
        b->step0 = 0; b->ticks = 0; b->drained = 0;
    }
    LH_LAUNCH(k_batch_reset_steps, dim3(1), dim3(64), 0, ctx->stream, b->rows_dev, b->sp_dev, b->B);
    LH_HIP(ctx, hipGetLastError());
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    b->sampling = true;
    return LH_OK;
}

int lh_batch_prompt(lh_batch* h, const uint32_t* const* prompts, const uint32_t* n_prompt, const float* x_in_dev, float* x_out_dev) {
    if (!h || !n_prompt) return LH_EINVAL;
    Batch* b = h->b;
    lh_ctx* ctx = b->ctx;
    const ModelDesc& m = b->pods[0]->md;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (m.first_stage() && !prompts) LH_FAIL(ctx, LH_EINVAL, "lh_batch_prompt: the first stage needs the prompts");
    if (!m.first_stage() && !x_in_dev) LH_FAIL(ctx, LH_EINVAL, "lh_batch_prompt: later stage needs the residual stream");
    if (!m.last_stage() && !x_out_dev) LH_FAIL(ctx, LH_EINVAL, "lh_batch_prompt: non-final stage needs an output buffer");
    uint32_t past[64];
    for (uint32_t i = 0; i < b->B; ++i) {   // everything is checked before anything runs
        if (!n_prompt[i] || n_prompt[i] > m.ctx) LH_FAIL(ctx, LH_EINVAL, "lh_batch_prompt: row %u: prompt of %u tokens outside 1..%u", i, n_prompt[i], m.ctx);
        if (m.first_stage()) {
            if (!prompts[i]) LH_FAIL(ctx, LH_EINVAL, "lh_batch_prompt: row %u has no prompt", i);
            for (uint32_t j = 0; j < n_prompt[i]; ++j)
                if (prompts[i][j] >= m.V) LH_FAIL(ctx, LH_EINVAL, "lh_batch_prompt: row %u: token id %u outside the vocabulary of %u", i, prompts[i][j], m.V);
        }
        past[i] = n_prompt[i];
    }
    int rc;
    LH_HIP(ctx, hipMemsetAsync(b->sp_dev, 0, sizeof(StepParams) * b->B, ctx->stream));   // sampling: the prompt's id is entry 0 of the row's output list
    size_t off = 0;   // rows of the pods, one after the other, in the residual stream buffers
    for (uint32_t i = 0; i < b->B; ++i) {
        Plan* p = b->pods[i];
        const uint32_t n = n_prompt[i];
        // the prompt as ONE Eval on the row's own plan (server.go:185-192).  (Plan 0 also holds the batch's scratch: it only ever grows, and
        // the captured tick is re-captured when it did - Plan::scratch_gen.)
        if ((rc = plan_eval(p, m.first_stage() ? prompts[i] : nullptr, x_in_dev ? x_in_dev + off * m.d : nullptr, x_out_dev ? x_out_dev + off * m.d : nullptr, n, 0, true))) return rc;
        if (m.last_stage()) {
            const float* last_row = n == 1 ? p->logits : p->logits + (size_t)(n - 1) * m.V;
            if (b->sampling) {
                if ((rc = sample_launch(ctx, last_row, m.V, b->ss_dev + i, b->ring_dev + (size_t)i * b->ring_cap, b->sp_dev + i, b->out_dev + (size_t)i * b->out_cap, b->ids_dev + i,
                                        nullptr, nullptr, nullptr, 1, b->smp_topk)))
                    return rc;
            } else {
                LH_LAUNCH(k_argmax_advance, dim3(1), dim3(1024), 0, ctx->stream, last_row, m.V, (StepParams*)nullptr, b->out_dev + (size_t)i * b->out_cap, b->ids_dev + i, 0);
                LH_HIP(ctx, hipMemcpyAsync(b->out_dev + (size_t)i * b->out_cap, b->ids_dev + i, 4, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        off += n;
    }
    if (m.first_stage() && m.last_stage()) LH_HIP(ctx, hipMemcpyAsync(b->tok_dev, b->ids_dev, (size_t)b->B * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if ((rc = batch_set(b, nullptr, past, 1))) return rc;   // every row now stands behind its prompt; out[row][0] holds the id the prompt produced
    LH_HIP(ctx, hipGetLastError());
    return LH_OK;
}

int lh_batch_stage(lh_batch* h, const float* x_in_dev, float* x_out_dev, float* logits_dev, uint32_t* ids_dev) {
    if (!h) return LH_EINVAL;
    Batch* b = h->b;
    lh_ctx* ctx = b->ctx;
    const ModelDesc& m = b->pods[0]->md;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (!m.first_stage() && !x_in_dev) LH_FAIL(ctx, LH_EINVAL, "lh_batch_stage: later stage needs the residual stream");
    if (!m.last_stage() && !x_out_dev) LH_FAIL(ctx, LH_EINVAL, "lh_batch_stage: non-final stage needs an output buffer");
    int rc;
    if ((rc = batch_tick(b, x_in_dev, x_out_dev))) return rc;
    if (m.last_stage() && logits_dev) LH_HIP(ctx, hipMemcpyAsync(logits_dev, b->logits(), (size_t)b->B * m.V * 4, hipMemcpyDeviceToDevice, ctx->stream));
    if (m.last_stage() && ids_dev) LH_HIP(ctx, hipMemcpyAsync(ids_dev, b->ids_dev, (size_t)b->B * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return LH_OK;
}

int lh_batch_decode(lh_batch* h, const uint32_t* first_tokens, const uint32_t* past, uint32_t n_steps, uint32_t* out_tokens, float* logits_last_host) {
    if (!h || !first_tokens || !past || !n_steps) return LH_EINVAL;
    Batch* b = h->b;
    lh_ctx* ctx = b->ctx;
    const ModelDesc& m = b->pods[0]->md;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (!m.first_stage() || !m.last_stage()) LH_FAIL(ctx, LH_EINVAL, "lh_batch_decode needs whole-model pods; use lh_batch_stage on a layer shard");
    int rc;
    if ((rc = batch_set(b, first_tokens, past))) return rc;
    b->collecting = true;
    b->gen.assign(b->B, std::vector<uint32_t>());
    for (uint32_t s = 0; s < n_steps && !rc; ++s) rc = batch_tick(b, nullptr, nullptr);   // (a row whose window is full swaps context first: batch_swap)
    if (!rc) rc = batch_drain(b);
    b->collecting = false;
    if (rc) return rc;
    if (out_tokens)
        for (uint32_t i = 0; i < b->B; ++i) memcpy(out_tokens + (size_t)i * n_steps, b->gen[i].data(), (size_t)n_steps * 4);
    if (logits_last_host) LH_HIP(ctx, hipMemcpyAsync(logits_last_host, b->logits(), (size_t)b->B * m.V * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_llama_profile_decode(lh_llama* m, uint32_t token, uint32_t past, uint32_t repeats, lh_kernel_time* out, uint32_t cap) {
    if (!m || !out || !repeats) return LH_EINVAL;
    lh_ctx* ctx = m->ctx;
    Plan* p = m->plan;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if ((rc = ensure_out_tokens(p, 1))) return rc;
    const float* pin = p->md.first_stage() ? nullptr : p->h;   // stage models: scratch stands in for the received residual
    float* pout = p->md.last_stage() ? nullptr : p->attn;
    ProfSink sink;
    sink.on = true;
    // warm-up (also sets kernel attributes), then timed eager steps at the same position
    g_prof = nullptr;
    if ((rc = upload_step_params(p, 0, token, past, 0))) return rc;
    if ((rc = enqueue_decode(p, p->sp_dev, pin, pout, false, nullptr))) return rc;
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    g_prof = &sink;
    LH_LAUNCH(k_park, dim3(1), dim3(1), 0, ctx->stream, (uint64_t)(repeats * 1500000ull));  // 15 ms per repeat of host queueing time
    for (uint32_t r = 0; r < repeats && !rc; ++r) rc = enqueue_decode(p, p->sp_dev, pin, pout, false, p->md.last_stage() ? p->argmax_dev : nullptr);
    g_prof = nullptr;
    hipStreamSynchronize(ctx->stream);
    std::vector<lh_kernel_time> acc;
    for (auto& r : sink.recs) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, r.e0, r.e1);
        hipEventDestroy(r.e0);
        hipEventDestroy(r.e1);
        size_t i = 0;
        for (; i < acc.size(); ++i) if (!strcmp(acc[i].name, r.name)) break;
        if (i == acc.size()) {
            lh_kernel_time t = {};
            snprintf(t.name, sizeof t.name, "%s", r.name);
            t.bytes_per_launch = r.bytes;
            acc.push_back(t);
        }
        acc[i].launches += 1;
        acc[i].total_ms += ms;
    }
    if (rc) return rc;
    // Second pass for the dominant weight-stream kernel: all of its launches of a step (one per layer, all weights distinct)
    // back to back between ONE event pair, queued behind the park kernel.  The per-kernel pairs above carry ~3 us of event
    // processing each; this is the figure that agrees with a rocprofv3 kernel trace (entry "<name>/b2b").
    size_t dom = acc.size();
    for (size_t i = 0; i < acc.size(); ++i)
        if (!strncmp(acc[i].name, "gemv_", 5) && (dom == acc.size() || acc[i].total_ms > acc[dom].total_ms)) dom = i;
    if (dom < acc.size() && acc[dom].launches >= repeats) {
        static thread_local char only_name[48];
        snprintf(only_name, sizeof only_name, "%s", acc[dom].name);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        g_only = only_name;
        LH_LAUNCH(k_park, dim3(1), dim3(1), 0, ctx->stream, (uint64_t)(repeats * 300000ull));
        hipEventRecord(e0, ctx->stream);
        for (uint32_t r = 0; r < repeats && !rc; ++r) rc = enqueue_decode(p, p->sp_dev, pin, pout, false, nullptr);
        hipEventRecord(e1, ctx->stream);
        g_only = nullptr;
        hipStreamSynchronize(ctx->stream);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        if (rc) return rc;
        lh_kernel_time t = {};
        snprintf(t.name, sizeof t.name, "%s/b2b", acc[dom].name);
        t.bytes_per_launch = acc[dom].bytes_per_launch;
        t.launches = acc[dom].launches;
        t.total_ms = ms;
        acc.push_back(t);
    }
    uint32_t n = (uint32_t)std::min<size_t>(acc.size(), cap);
    for (uint32_t i = 0; i < n; ++i) out[i] = acc[i];
    return (int)n;
}

}  // extern "C"
