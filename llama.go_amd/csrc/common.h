// csrc/common.h — internal state of libllamahip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <unordered_map>
#include <mutex>
#include <atomic>
#include <memory>
#include "../../include/llamahip.h"

namespace lh {

void set_error(lh_ctx* ctx, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define LH_HIP(ctx, expr)                                                                               \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess) {                                                                        \
            lh::set_error(ctx, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return LH_EHIP;                                                                             \
        }                                                                                               \
    } while (0)

// Every kernel launch of the library goes through LH_LAUNCH: with the route log on (lh_route_log; tests/test_gpu_zz_routes.py) the kernel FAMILY (the
// __global__'s name without its template arguments) is noted, so that the test suite can assert that every product kernel is reached by a parity test.
extern std::atomic<int> g_route_log_on;
void route_note(const char* kernel_expr);
#define LH_LAUNCH(kern, ...)                                                        \
    do {                                                                            \
        if (lh::g_route_log_on.load(std::memory_order_relaxed)) lh::route_note(#kern); \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                                      \
    } while (0)
#define LH_LAUNCH_AS(family, kern, ...)                                             \
    do {                                                                            \
        if (lh::g_route_log_on.load(std::memory_order_relaxed)) lh::route_note(family); \
        hipLaunchKernelGGL(kern, __VA_ARGS__);                                      \
    } while (0)

#define LH_FAIL(ctx, code, ...)          \
    do {                                 \
        lh::set_error(ctx, __VA_ARGS__); \
        return code;                     \
    } while (0)

// A persistent device buffer (weights / KV cache), shared by every context of the device.
struct Buffer {
    float* dev = nullptr;
    uint64_t nfloats = 0;  // f32 elements (for block-int8: logical elements)
    uint64_t bytes = 0;
    int dtype = 0;              // ml.DType: 0 = f32, 7 = block-int8 (planes: int8 quants at dev, fp32 scales at `scales`)
    float* scales = nullptr;    // block-int8: [rows][cols/32]
    uint32_t rows = 0, cols = 0;
    uint64_t key = 0;
    int device = 0;
    // a KV cache: the token evaluated at every position, as far as the host knows it (0xFFFFFFFF = unknown).  Lives with the BUFFER so that
    // every plan over the cache (the graph path's and the resident loop's) sees the same history, and a new cache starts without one;
    // created on first use (plan.hip: context swap of the generation loops, server.go:160-172).
    std::shared_ptr<std::vector<uint32_t>> kv_hist;
};

// Per-device shared state: buffer registry (Model is shared read-only across pods, server.go:45).
struct DeviceState {
    int device = 0;
    hipDeviceProp_t prop;
    int num_cu = 0;
    std::mutex mu;
    std::unordered_map<lh_buf, std::unique_ptr<Buffer>> bufs;
    std::unordered_map<uint64_t, lh_buf> by_key;
    lh_buf next_id = 1;
    std::atomic<uint64_t> bufs_gen{1};   // bumped (under mu) by every insertion into / removal from bufs: contexts keep lock-free snapshots (find_buffer_fast)
    // RoPE tables, one per rotation width `dims`: [positions][dims/2] of (cos, sin) in f64, built on the host with libm
    // exactly as the reference computes them per element (ml.go:2307-2310).  A table that has to grow is REPLACED by a larger
    // one and the old allocation is kept alive (captured graphs and plans of other contexts hold its address; tables are
    // small), so a pointer handed out by ensure_rope_table stays valid for the positions it was asked for.
    struct RopeTable { double2* dev = nullptr; uint32_t positions = 0; };
    std::unordered_map<uint32_t, RopeTable> rope;
};
DeviceState* device_state(int device);
Buffer* find_buffer(DeviceState* ds, lh_buf id);
Buffer* find_buffer_fast(lh_ctx* ctx, lh_buf id);   // the same through the context's snapshot of the table: no lock, no hashing (ids are small integers)

struct Plan;  // fused LLaMA plan (plan.hip)

}  // namespace lh

struct lh_ctx {
    int device = 0;
    lh::DeviceState* ds = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // per-graph scratch arena (reset on every lh_graph_compute; grows, never shrinks)
    char* arena = nullptr;
    uint64_t arena_bytes = 0;
    // pinned staging for small host leafs / parameters
    char* staging = nullptr;
    uint64_t staging_bytes = 0;
    // snapshot of ds->bufs indexed by buffer id, valid while buf_snap_gen == ds->bufs_gen (a Matcher run looks up ~300 weight buffers per Eval)
    std::vector<lh::Buffer*> buf_snap;
    uint64_t buf_snap_gen = 0;
    // the rows a fused Eval's caller is known to read (LH_GRAPH_LAST_ROW_LOGITS: the last logits row, llama.go:394-401) are copied to pinned host
    // memory behind the Eval's kernels, in front of its ONE synchronisation; lh_node_read of exactly that range is then a host copy
    float* out_pinned = nullptr;
    uint64_t out_pinned_floats = 0;
    int64_t pre_index = -1;
    uint64_t pre_off = 0, pre_n = 0;
    // last computed graph: device address + element count of every tensor (for lh_node_read)
    std::vector<float*> last_ptr;
    std::vector<uint64_t> last_len;
    int last_fused = 0;
    // fused plans cached by structural signature
    std::vector<lh::Plan*> plans;
    // split-K partial products of the prefill GEMM (short prompts); grows, never shrinks.  splitk_gen counts the re-allocations: a
    // captured graph that holds the address (lh_batch ticks) is re-captured when it changed
    float* splitk = nullptr;
    uint64_t splitk_floats = 0;
    uint64_t splitk_gen = 0;
    // the activation rows of a long-prompt GEMM as three bf16 planes (k_gemm_b9); grows, never shrinks
    uint16_t* xs3 = nullptr;
    uint64_t xs3_elems = 0;
    // lh_ctx_time_computes: events around each lh_graph_compute + host time inside it
    bool tc_on = false, tc_end = false;
    hipEvent_t tc_ev0 = nullptr, tc_ev1 = nullptr;
    uint64_t tc_calls = 0;
    double tc_wall_us = 0, tc_dev_us = 0;
};

namespace lh {
int ensure_arena(lh_ctx* ctx, uint64_t bytes);
int ensure_staging(lh_ctx* ctx, uint64_t bytes);
int ensure_rope_table(lh_ctx* ctx, uint32_t positions, uint32_t dims, const double2** table);
inline int select_device(lh_ctx* ctx) { return hipSetDevice(ctx->device) == hipSuccess ? 0 : LH_EHIP; }
}  // namespace lh
