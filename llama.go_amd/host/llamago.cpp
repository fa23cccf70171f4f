// host/llamago.cpp — C++ mirror of the reference's pkg/ml + pkg/llama operator surface (include/llamago.h)
// on top of the C-ABI boundary (include/llamahip.h).  This file is what the Go side of a llama.go build with
// the `hip` tag does (INTEGRATION.md / go/ml_hip.go): graphs are BUILT on the host exactly like the reference
// (shapes, strides, view aliasing, post-order DFS), and ml_GraphCompute crosses the boundary ONCE per Eval.
// No arithmetic of the hot path happens here and there is no CPU fallback: without a GPU every compute call fails.
//
// The reference is Go; the toolchain is absent from this image, hence C++ (prompt rule ②).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>
#include <unordered_map>
#include <unordered_set>
#include "../../include/llamago.h"
#include "../../include/llamahip.h"
#include "../../include/llamago_ext.h"

#define MAX_NODES 4096  // ml.go:20

static thread_local std::string g_err;
static void* halt(const char* msg) { g_err = msg; return nullptr; }
static int halt_rc(const std::string& msg) { g_err = msg; return 1; }
extern "C" const char* ml_LastError(void) { return g_err.c_str(); }

// ---- process-wide device selection -----------------------------------------------------------------
static void* g_stream_for_new_contexts = nullptr;
static int device_index() {
    const char* e = getenv("LLAMAGO_DEVICE");
    if (e && *e) return atoi(e);
    e = getenv("LOCAL_RANK");
    if (e && *e) return atoi(e);
    return 0;
}
static lh_ctx* g_model_ctx = nullptr;  // context used for model-level work (weight registration, fills)
static lh_ctx* model_ctx() {
    if (!g_model_ctx) {
        int rc = lh_ctx_create(device_index(), nullptr, &g_model_ctx);
        if (rc) { g_err = std::string("no HIP context: ") + lh_last_error(nullptr); return nullptr; }
    }
    return g_model_ctx;
}

struct ml_context {  // ml.Context ml.go:50-57
    int maxThreads;
    lh_ctx* hip;
    uint64_t generation = 0;  // bumped by every GraphCompute (validity of tensor->graph indices)
    // scratch of graph_compute, kept between calls (an Eval per token flattens ~1500 tensors each time): the array the C side sees, the owner
    // list, and the index of every persistent (shared, read-only) tensor by its device buffer id, stamped with the call that set it
    std::vector<lh_tensor> flat;
    std::vector<ml_tensor*> flat_leafs;
    struct SharedIdx { uint64_t mark; int idx; };
    std::vector<SharedIdx> shared_idx;
};

struct ml_tensor {  // ml.Tensor ml.go:180-203
    int type = 0;
    uint32_t dims = 0;
    uint32_t ne[4] = {1, 1, 1, 1};
    uint32_t nb[4] = {4, 4, 4, 4};
    int op = ML_OP_NONE;
    ml_tensor *src0 = nullptr, *src1 = nullptr;
    float* data = nullptr;  // host Data (NULL for device-only persistent tensors)
    bool owns = false;
    // aliasing made explicit for the C side: the tensor whose allocation these bytes live in + float offset
    ml_tensor* base = nullptr;
    uint64_t base_off = 0;
    // device residency of persistent leafs (weights, KV cache)
    lh_buf buf = 0;
    bool persistent = false;
    // position in the last computed graph
    ml_context* last_ctx = nullptr;
    uint64_t last_gen = 0;
    uint32_t last_index = 0;
    bool want_output = false;  // the host reads this node's Data after GraphCompute (-> LH_T_OUTPUT)
    ml_tensor *gc_next = nullptr, *gc_prev = nullptr;  // per-thread list of constructor-made tensors (see ml_FreeGraph)
    bool gc_owned = false;
    // scratch marks of graph construction / flattening (valid when the generation matches): no hash containers on the Eval path
    uint64_t visit_gen = 0, flat_gen = 0;
    int flat_idx = 0;
};

struct ml_graph {  // ml.Graph ml.go:31-45
    std::vector<ml_tensor*> nodes, leafs;
    // "already visited": per-Eval tensors (owned by the building thread) carry a generation mark; persistent tensors (weights, KV
    // caches) are shared read-only by every pod's goroutine (server.go:45) and must not be written, so they go through a set.
    // Same visit order as the reference's linear scans (ml.go:657-668) without the O(n^2).
    uint64_t gen = 0;
    std::unordered_set<const ml_tensor*> seen_shared;
};
static uint64_t g_mark_counter = 0;  // graphs are built and flattened under the caller's serialisation (one ml.Context per goroutine)
static std::mutex g_mark_mu;
static uint64_t next_mark() { std::lock_guard<std::mutex> lk(g_mark_mu); return ++g_mark_counter; }

static thread_local ml_tensor* g_gc_head = nullptr;
static thread_local std::vector<ml_tensor*> g_pool;  // recycled tensor records: an Eval creates ~1600 and frees them all
static thread_local bool g_gc_enabled = true;

static uint64_t nelements(const ml_tensor* t) { return (uint64_t)t->ne[0] * t->ne[1] * t->ne[2] * t->ne[3]; }
static bool same_shape(const ml_tensor* a, const ml_tensor* b) { return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3]; }

// NewTensor ml.go:760-783 — strides always re-derived contiguous; data == alias of `alias_of` when given
static ml_tensor* new_tensor(int dt, uint32_t dims, uint32_t ne0, uint32_t ne1, uint32_t ne2, uint32_t ne3, ml_tensor* alias_of, uint64_t extra_off,
                             bool alloc_host = true) {
    ml_tensor* t;
    if (!g_pool.empty()) { t = g_pool.back(); g_pool.pop_back(); *t = ml_tensor(); } else t = new ml_tensor();
    t->type = dt;
    t->dims = dims;
    t->ne[0] = ne0; t->ne[1] = ne1; t->ne[2] = ne2; t->ne[3] = ne3;
    t->nb[0] = 4; t->nb[1] = ne0 * 4; t->nb[2] = ne0 * ne1 * 4; t->nb[3] = ne0 * ne1 * ne2 * 4;
    if (alias_of) {
        t->base = alias_of->base;
        t->base_off = alias_of->base_off + extra_off;
        t->data = alias_of->base->data ? alias_of->base->data + t->base_off : nullptr;
    } else {
        t->base = t;
        if (alloc_host) {
            const uint64_t n = nelements(t);
            t->data = (float*)calloc(n ? n : 1, sizeof(float));
            t->owns = true;
        }
    }
    if (g_gc_enabled) { t->gc_owned = true; t->gc_next = g_gc_head; if (g_gc_head) g_gc_head->gc_prev = t; g_gc_head = t; }
    return t;
}
static void gc_unlink(ml_tensor* t) {
    if (!t->gc_owned) return;
    if (t->gc_prev) t->gc_prev->gc_next = t->gc_next; else if (g_gc_head == t) g_gc_head = t->gc_next;
    if (t->gc_next) t->gc_next->gc_prev = t->gc_prev;
    t->gc_owned = false; t->gc_next = t->gc_prev = nullptr;
}
static ml_tensor* new_leaf(int dt, uint32_t dims, uint32_t ne0, uint32_t ne1, uint32_t ne2, bool alloc_host = true) {
    const bool save = g_gc_enabled;
    g_gc_enabled = false;
    ml_tensor* t = new_tensor(dt, dims, ne0, ne1, ne2, 1, nullptr, 0, alloc_host);
    g_gc_enabled = save;
    return t;
}
static ml_tensor* view_tensor(ml_tensor* s) { return new_tensor(s->type, s->dims, s->ne[0], s->ne[1], s->ne[2], s->ne[3], s, 0); }  // ml.go:231
// DupTensor ml.go:236 — a fresh result tensor.  Results live in HBM only: no host Data is allocated for them (the
// reference's per-node make([]float32) + forced GC is exactly the churn SURVEY §8a row 21 lists); read them with ml_TensorRead.
static ml_tensor* dup_tensor(ml_tensor* s) { return new_tensor(s->type, s->dims, s->ne[0], s->ne[1], s->ne[2], s->ne[3], nullptr, 0, false); }
static ml_tensor* node2(ml_tensor* r, int op, ml_tensor* a, ml_tensor* b) { r->op = op; r->src0 = a; r->src1 = b; return r; }
static void free_tensor(ml_tensor* t) {
    if (!t) return;
    if (t->owns) free(t->data);
    if (t->buf && t->persistent && g_model_ctx) lh_buf_free(g_model_ctx, t->buf);
    if (g_pool.size() < 8192) g_pool.push_back(t); else delete t;
}

extern "C" {

// ---- context ---------------------------------------------------------------------------------------
ml_context* ml_NewContext(int maxThreads, int useAVX, int useNEON) {  // ml.go:59-74
    (void)useAVX; (void)useNEON;  // CPU kernel selectors of the reference (ml.go:52-53); this backend always runs HIP
    lh_ctx* h = nullptr;
    if (lh_ctx_create(device_index(), g_stream_for_new_contexts, &h)) return (ml_context*)halt(lh_last_error(nullptr));
    ml_context* c = new ml_context();
    c->maxThreads = maxThreads;
    c->hip = h;
    return c;
}
void ml_ReleaseContext(ml_context* ctx) {  // ml.go:77-80
    if (!ctx) return;
    lh_ctx_destroy(ctx->hip);
    delete ctx;
}

// ---- tensors ---------------------------------------------------------------------------------------
ml_tensor* ml_NewTensor1D(ml_context*, int dt, uint32_t ne0) { return new_leaf(dt, 1, ne0, 1, 1); }
ml_tensor* ml_NewTensor2D(ml_context*, int dt, uint32_t ne0, uint32_t ne1) { return new_leaf(dt, 2, ne0, ne1, 1); }
ml_tensor* ml_NewTensor3D(ml_context*, int dt, uint32_t ne0, uint32_t ne1, uint32_t ne2) { return new_leaf(dt, 3, ne0, ne1, ne2); }
ml_tensor* ml_NewFP32(ml_context* ctx, float v) { ml_tensor* t = ml_NewTensor1D(ctx, ML_TYPE_F32, 1); t->data[0] = v; return t; }  // ml.go:915-930
float* ml_TensorData(ml_tensor* t) { return t->data; }
void ml_TensorShape(const ml_tensor* t, uint32_t ne[4], uint32_t nb[4]) { for (int i = 0; i < 4; i++) { ne[i] = t->ne[i]; nb[i] = t->nb[i]; } }
int ml_TensorOp(const ml_tensor* t) { return t->op; }
uint64_t ml_Nelements(const ml_tensor* t) { return nelements(t); }
void ml_TensorDirty(ml_tensor* t) { (void)t; /* non-persistent leafs are uploaded on every GraphCompute */ }
void ml_FreeTensor(ml_tensor* t) { free_tensor(t); }

int ml_TensorRead(ml_context* ctx, ml_tensor* t, float* dst, uint64_t n) {
    if (t->base->persistent && t->base->buf) {  // weights / KV cache: straight from HBM
        lh_ctx* h = ctx ? ctx->hip : model_ctx();
        if (!h) return 1;
        if (lh_buf_read(h, t->base->buf, t->base_off, dst, n)) return halt_rc(lh_last_error(h));
        return 0;
    }
    if (t->op == ML_OP_NONE && t->data) { memcpy(dst, t->data, n * 4); return 0; }  // plain host leaf
    if (!ctx || t->last_ctx != ctx || t->last_gen != ctx->generation) return halt_rc("ml_TensorRead: tensor is not part of the last computed graph");
    if (lh_node_read(ctx->hip, t->last_index, 0, dst, n)) return halt_rc(lh_last_error(ctx->hip));
    return 0;
}

// ---- operator constructors (shapes/strides/views exactly as the reference builds them) -------------------
ml_tensor* ml_Mul(ml_context*, ml_tensor* a, ml_tensor* b) {  // ml.go:241-287
    if (!same_shape(a, b)) return (ml_tensor*)halt("[STOP] MulImpl - tensors of different shapes!");
    return node2(dup_tensor(a), ML_OP_MUL, a, b);
}
ml_tensor* ml_Add(ml_context*, ml_tensor* a, ml_tensor* b) { return node2(dup_tensor(a), ML_OP_ADD, a, b); }  // ml.go:321-360
ml_tensor* ml_MulMat(ml_context*, ml_tensor* a, ml_tensor* b) {  // ml.go:295-318
    ml_tensor* r = new_tensor(ML_TYPE_F32, a->dims < b->dims ? a->dims : b->dims, a->ne[1], b->ne[1], a->ne[2], b->ne[3], nullptr, 0, false);
    return node2(r, ML_OP_MUL_MAT, a, b);
}
ml_tensor* ml_Repeat(ml_context*, ml_tensor* a, ml_tensor* b) {  // ml.go:487-513
    if (same_shape(a, b)) return a;
    return node2(new_tensor(a->type, b->dims, b->ne[0], b->ne[1], b->ne[2], b->ne[3], nullptr, 0, false), ML_OP_REPEAT, a, b);
}
ml_tensor* ml_GetRows(ml_context*, ml_tensor* a, ml_tensor* b) {  // ml.go:528-557
    return node2(new_tensor(ML_TYPE_F32, 2, a->ne[0], b->ne[0], 1, 1, nullptr, 0, false), ML_OP_GET_ROWS, a, b);
}
ml_tensor* ml_RMSNorm(ml_context*, ml_tensor* a) { return node2(dup_tensor(a), ML_OP_RMS_NORM, a, nullptr); }  // ml.go:559-597
ml_tensor* ml_View1D(ml_context*, ml_tensor* a, uint32_t ne0, uint32_t offset) {  // ml.go:601-617 (offset in floats)
    if (a->base_off + (uint64_t)offset + ne0 > nelements(a->base)) return (ml_tensor*)halt("[HALT] View1D : slice bounds out of range");
    return node2(new_tensor(a->type, 1, ne0, 1, 1, 1, a, offset), ML_OP_VIEW, a, nullptr);
}
ml_tensor* ml_Copy(ml_context*, ml_tensor* a, ml_tensor* b) { return node2(view_tensor(b), ML_OP_CPY, a, b); }  // ml.go:700-735
ml_tensor* ml_Permute(ml_context*, ml_tensor* a, uint32_t ax0, uint32_t ax1, uint32_t ax2, uint32_t ax3) {  // ml.go:786-845
    if (ax0 > 3 || ax1 > 3 || ax2 > 3 || ax3 > 3) return (ml_tensor*)halt("[STOP] Permute error");
    ml_tensor* r = view_tensor(a);
    uint32_t ne[4], nb[4];
    ne[ax0] = a->ne[0]; ne[ax1] = a->ne[1]; ne[ax2] = a->ne[2]; ne[ax3] = a->ne[3];
    nb[ax0] = a->nb[0]; nb[ax1] = a->nb[1]; nb[ax2] = a->nb[2]; nb[ax3] = a->nb[3];
    for (int i = 0; i < 4; i++) { r->ne[i] = ne[i]; r->nb[i] = nb[i]; }
    return node2(r, ML_OP_PERMUTE, a, nullptr);
}
ml_tensor* ml_Rope(ml_context*, ml_tensor* a, uint32_t past, uint32_t dims, uint32_t mode) {  // ml.go:848-880
    ml_tensor* r = view_tensor(a);
    ml_tensor* b = new_tensor(ML_TYPE_I32, 1, 3, 1, 1, 1, nullptr, 0);
    b->data[0] = (float)past; b->data[1] = (float)dims; b->data[2] = (float)mode;
    return node2(r, ML_OP_ROPE, a, b);
}
ml_tensor* ml_Reshape3D(ml_context*, ml_tensor* a, uint32_t ne0, uint32_t ne1, uint32_t ne2) {  // ml.go:882-912
    return node2(new_tensor(a->type, 3, ne0, ne1, ne2, 1, a, 0), ML_OP_RESHAPE, a, nullptr);
}
ml_tensor* ml_Scale(ml_context*, ml_tensor* a, ml_tensor* b) { return node2(view_tensor(a), ML_OP_SCALE, a, b); }  // ml.go:933-961
ml_tensor* ml_DiagMaskInf(ml_context*, ml_tensor* a, uint32_t past) {  // ml.go:968-990
    ml_tensor* b = new_tensor(ML_TYPE_F32, 1, 1, 1, 1, 1, nullptr, 0);
    b->data[0] = (float)past;
    return node2(view_tensor(a), ML_OP_DIAG_MASK_INF, a, b);
}
ml_tensor* ml_SoftMax(ml_context*, ml_tensor* a) { return node2(view_tensor(a), ML_OP_SOFT_MAX, a, nullptr); }  // ml.go:993-1014
ml_tensor* ml_Silu(ml_context*, ml_tensor* a) { return node2(dup_tensor(a), ML_OP_SILU, a, nullptr); }         // ml.go:1018-1043

// ---- graph  ml.go:619-697 ------------------------------------------------------------------------------
ml_graph* ml_NewGraph(void) { ml_graph* g = new ml_graph(); g->gen = next_mark(); g->nodes.reserve(2048); g->leafs.reserve(1024); return g; }
// Go's GC reclaims the per-Eval tensors; here a graph owns what it REACHED: its nodes, and the leafs the operator constructors made
// on the way (Rope / DiagMaskInf parameters, Copy destinations).  Leafs made with ml_NewTensor* belong to the caller, weights and KV
// caches to their model / context.  Tensors of OTHER graphs built on the same thread are left alone (a node shared by two graphs goes
// with the first one freed).  Constructor-made tensors that never reached a graph stay on the thread's list until
// llamago_CollectGarbage (the runtime.GC() of llama.go:423) or the failure path of llama_Eval releases them.
static void free_owned(ml_tensor* t) { if (t->gc_owned) { gc_unlink(t); free_tensor(t); } }
void ml_FreeGraph(ml_graph* g) {
    if (!g) return;
    for (ml_tensor* t : g->nodes) free_owned(t);
    for (ml_tensor* t : g->leafs) free_owned(t);
    delete g;
}
static void gc_free_down_to(ml_tensor* mark) {  // everything this thread's constructors made after `mark` was the list head
    while (g_gc_head && g_gc_head != mark) { ml_tensor* t = g_gc_head; gc_unlink(t); free_tensor(t); }
}
void llamago_CollectGarbage(void) { gc_free_down_to(nullptr); }
static int visit_parents(ml_graph* g, ml_tensor* node) {  // ml.go:647-697
    if (node->persistent) {
        if (!g->seen_shared.insert(node).second) return 0;
    } else {
        if (node->visit_gen == g->gen) return 0;
        node->visit_gen = g->gen;
    }
    if (node->src0 && visit_parents(g, node->src0)) return 1;
    if (node->src1 && visit_parents(g, node->src1)) return 1;
    if (node->op == ML_OP_NONE) {
        if (g->leafs.size() >= MAX_NODES) return halt_rc("[HALT] graph: too many leafs");
        g->leafs.push_back(node);
    } else {
        if (g->nodes.size() >= MAX_NODES) return halt_rc("[HALT] graph: too many nodes");
        g->nodes.push_back(node);
    }
    return 0;
}
int ml_BuildForwardExpand(ml_graph* g, ml_tensor* t) {  // ml.go:620-644
    const size_t n0 = g->nodes.size();
    if (visit_parents(g, t)) return 1;
    if (g->nodes.size() > n0 && g->nodes.back() != t) return halt_rc("[STOP] BuildForwardImpl : the last added node should always be starting point!");
    return 0;
}
uint32_t ml_GraphNodesCount(const ml_graph* g) { return (uint32_t)g->nodes.size(); }
ml_tensor* ml_GraphNode(const ml_graph* g, uint32_t i) { return i < g->nodes.size() ? g->nodes[i] : nullptr; }

// The array the C side sees: leafs, the storage owners that are not in the graph themselves, then the nodes in the order ml.GraphCompute
// walks them (ml.go:1501-1526).  Lands in ctx->flat (scratch kept between calls); *nl / *nn = leaf and node counts.
static void flatten_graph(ml_context* ctx, ml_graph* g, uint32_t* nl_out, uint32_t* nn_out) {
    // every storage owner must be part of the array; indices live in the tensors (flat_gen / flat_idx)
    std::vector<ml_tensor*>& leafs = ctx->flat_leafs;
    leafs.assign(g->leafs.begin(), g->leafs.end());
    const uint64_t fg = next_mark();
    // persistent (shared, read-only) tensors must not be written (see ml_graph): their index lives in a table of this context, by buffer id
    auto& sidx = ctx->shared_idx;
    auto slot = [&](const ml_tensor* t) -> ml_context::SharedIdx& {
        if ((size_t)t->buf >= sidx.size()) sidx.resize((size_t)t->buf + 64, ml_context::SharedIdx{0, -1});
        return sidx[(size_t)t->buf];
    };
    auto set_idx = [&](ml_tensor* t, int i) { if (t->persistent) slot(t) = ml_context::SharedIdx{fg, i}; else { t->flat_gen = fg; t->flat_idx = i; } };
    auto has_idx = [&](const ml_tensor* t) { return t->persistent ? slot(t).mark == fg : t->flat_gen == fg; };
    auto index_of = [&](const ml_tensor* t) { return t->persistent ? slot(t).idx : t->flat_idx; };
    for (size_t i = 0; i < leafs.size(); ++i) set_idx(leafs[i], (int)i);
    for (ml_tensor* t : g->nodes) set_idx(t, -2);
    auto add_owner = [&](ml_tensor* t) {
        ml_tensor* b = t->base;
        if (!has_idx(b)) { set_idx(b, (int)leafs.size()); leafs.push_back(b); }
    };
    for (ml_tensor* t : g->leafs) add_owner(t);
    for (ml_tensor* t : g->nodes) add_owner(t);
    const uint32_t nl = (uint32_t)leafs.size(), nn = (uint32_t)g->nodes.size();
    for (uint32_t i = 0; i < nn; ++i) set_idx(g->nodes[i], (int)(nl + i));
    std::vector<lh_tensor>& T = ctx->flat;
    T.resize(nl + nn);
    ctx->generation++;
    for (uint32_t i = 0; i < nl + nn; ++i) {
        ml_tensor* t = i < nl ? leafs[i] : g->nodes[i - nl];
        lh_tensor& o = T[i];
        memset(&o, 0, sizeof o);
        o.op = (uint8_t)t->op;
        o.flags = t->want_output ? LH_T_OUTPUT : 0;
        o.dtype = (uint8_t)(t->type == ML_TYPE_I32 ? ML_TYPE_F32 : t->type);  // "I32" parameter tensors hold fp32 (ml.go:864-867)
        for (int k = 0; k < 4; ++k) { o.ne[k] = t->ne[k]; o.nb[k] = t->nb[k]; }
        o.src0 = t->src0 ? index_of(t->src0) : -1;
        o.src1 = t->src1 ? index_of(t->src1) : -1;
        o.storage = index_of(t->base);
        o.view_off = t->base_off;
        if (t->base == t) {
            o.buf = t->persistent ? t->buf : 0;
            o.host = (t->op == ML_OP_NONE && !t->persistent) ? t->data : nullptr;
        }
        if (!t->persistent) {  // shared tensors stay read-only (they are read back through their buffer, ml_TensorRead)
            t->last_ctx = ctx;
            t->last_gen = ctx->generation;
            t->last_index = i;
        }
    }
    *nl_out = nl; *nn_out = nn;
}
// ml.GraphCompute ml.go:1411-1528 -> ONE call across the C-ABI.
static int graph_compute(ml_context* ctx, ml_graph* g, uint32_t flags) {
    g_err.clear();
    if (!ctx || !ctx->hip) return halt_rc("ml_GraphCompute: no HIP context");
    const auto tf0 = std::chrono::steady_clock::now();
    uint32_t nl = 0, nn = 0;
    flatten_graph(ctx, g, &nl, &nn);
    static const bool timing = getenv("LLAMAGO_TIMING") != nullptr;
    if (timing) fprintf(stderr, "[llamago] flatten %u tensors: %.1f us\n", nl + nn, (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - tf0).count() / 1000.0);
    if (lh_graph_compute(ctx->hip, ctx->flat.data(), nl, nn, flags)) return halt_rc(lh_last_error(ctx->hip));
    return 0;
}
int ml_GraphCompute(ml_context* ctx, ml_graph* g) {
    const char* e = getenv("LLAMAGO_NO_FUSION");  // debugging aid: force the node-by-node path
    return graph_compute(ctx, g, (e && e[0] == '1') ? LH_GRAPH_NO_FUSION : 0);
}
// product extensions (not in the reference): node-by-node execution, and which path the last graph took
int llamago_GraphComputeNoFusion(ml_context* ctx, ml_graph* g) { return graph_compute(ctx, g, LH_GRAPH_NO_FUSION); }
int llamago_LastGraphFused(ml_context* ctx) { return ctx && ctx->hip ? lh_last_graph_fused(ctx->hip) : 0; }
void llamago_SetStream(void* hip_stream) { g_stream_for_new_contexts = hip_stream; }
int llamago_DeviceCount(void) { return lh_device_count(); }
int llamago_HbmReadProbe(uint64_t bytes, uint32_t repeats, float* gbps) {
    lh_ctx* h = model_ctx();
    if (!h) return 1;
    if (lh_hbm_read_probe(h, bytes, repeats, gbps)) return halt_rc(lh_last_error(h));
    return 0;
}

}  // extern "C"

// ======================================================================================================
// pkg/llama
// ======================================================================================================
struct llama_layer { ml_tensor *attentionNorm, *wq, *wk, *wv, *wo, *ffn_norm, *w1, *w2, *w3; };  // llama.go:128-146
struct llama_model {  // llama.go:181-193
    llama_hparams hp;
    uint32_t ffSize;
    int wtype = ML_TYPE_F32;  // dtype of the weight matrices (ML_TYPE_Q8_0 after llamago_QuantizeModelQ8)
    ml_tensor *tokEmbeddings = nullptr, *norm = nullptr, *output = nullptr;
    std::vector<llama_layer> layers;
    uint32_t layer0, layer1;
    // contexts / pipelines alive on this model: their plans hold raw device addresses of the weights, so the weights may neither be
    // re-quantised nor released under them (Go keeps the Model reachable through every Context)
    std::mutex mu;
    int users = 0;
    bool release_pending = false;
};
static void free_model_now(llama_model* m);
static void model_acquire(llama_model* m) { std::lock_guard<std::mutex> lk(m->mu); m->users++; }
static void model_release(llama_model* m) {
    bool last;
    { std::lock_guard<std::mutex> lk(m->mu); last = --m->users == 0 && m->release_pending; }
    if (last) free_model_now(m);
}
struct llama_context {  // llama.go:83-88
    ml_tensor *K, *V;
    std::vector<float> logits;
    ml_context* mlctx;
    llama_model* model;
    uint32_t ctxSize;
    uint32_t keepCount = 0;        // ModelParams.KeepCount llama.go:47
    std::vector<float> embedding;  // lctx.Embedding (llama.go:88; allocated when ModelParams.Embedding is set, llama.go:414-419)
    lh_llama* resident = nullptr;  // plan handle for the device-resident loop / stages (created on demand)
    bool holds_model = false;
    struct eval_cache* decode_graph = nullptr;   // the graph of a one-token Eval, kept between calls (see eval_cache)
};

static void eval_cache_free(struct eval_cache* c);
static uint32_t ff_size(uint32_t embd, uint32_t mult) { return ((2 * (4 * embd) / 3 + mult - 1) / mult) * mult; }  // llama.go:761

// persistent device tensor without a host copy
static ml_tensor* new_weight(uint32_t dims, uint32_t ne0, uint32_t ne1) {
    lh_ctx* h = model_ctx();
    if (!h) return nullptr;
    ml_tensor* t = new_leaf(ML_TYPE_F32, dims, ne0, ne1, 1, /*alloc_host=*/false);
    t->persistent = true;
    const uint32_t ne[4] = {ne0, ne1, 1, 1};
    if (lh_tensor_register(h, 0, 0, ne, 1, nullptr, &t->buf)) { g_err = lh_last_error(h); delete t; return nullptr; }
    return t;
}

static llama_model* alloc_model(const llama_hparams* hp, uint32_t layer0, uint32_t layer1) {  // llama.go:819-863
    llama_model* m = new llama_model();
    m->hp = *hp;
    m->ffSize = ff_size(hp->embdSize, hp->multSize);
    if (layer1 == 0 || layer1 > hp->layersCount) layer1 = hp->layersCount;
    m->layer0 = layer0; m->layer1 = layer1;
    const uint32_t d = hp->embdSize, V = hp->vocabSize, F = m->ffSize;
    bool ok = true;
    if (layer0 == 0) ok &= (m->tokEmbeddings = new_weight(2, d, V)) != nullptr;
    if (layer1 == hp->layersCount) {
        ok &= (m->norm = new_weight(1, d, 1)) != nullptr;
        ok &= (m->output = new_weight(2, d, V)) != nullptr;
    }
    m->layers.assign(hp->layersCount, llama_layer{});
    for (uint32_t i = layer0; i < layer1 && ok; i++) {
        llama_layer& l = m->layers[i];
        ok &= (l.attentionNorm = new_weight(1, d, 1)) != nullptr;
        ok &= (l.wq = new_weight(2, d, d)) != nullptr;
        ok &= (l.wk = new_weight(2, d, d)) != nullptr;
        ok &= (l.wv = new_weight(2, d, d)) != nullptr;
        ok &= (l.wo = new_weight(2, d, d)) != nullptr;
        ok &= (l.ffn_norm = new_weight(1, d, 1)) != nullptr;
        ok &= (l.w1 = new_weight(2, d, F)) != nullptr;
        ok &= (l.w2 = new_weight(2, F, d)) != nullptr;
        ok &= (l.w3 = new_weight(2, d, F)) != nullptr;
    }
    if (!ok) { llama_FreeModel(m); return nullptr; }
    return m;
}

enum { TID_TOK = 0, TID_NORM = 1, TID_OUT = 2, TID_LAYER0 = 16, TID_PER_LAYER = 16 };

static int fill(ml_tensor* t, uint64_t seed, uint32_t tid, float scale, float offset) {
    if (!t) return 0;
    lh_ctx* h = model_ctx();
    if (lh_buf_fill_synth(h, t->buf, 0, nelements(t), seed, tid, scale, offset)) return halt_rc(lh_last_error(h));
    return 0;
}

extern "C" {

ml_tensor* llama_ModelTensor(llama_model* m, const char* name) {  // model.tensors llama.go:826-861
    if (!strcmp(name, "tok_embeddings.weight")) return m->tokEmbeddings;
    if (!strcmp(name, "norm.weight")) return m->norm;
    if (!strcmp(name, "output.weight")) return m->output;
    unsigned i;
    char rest[64];
    if (sscanf(name, "layers.%u.%63s", &i, rest) == 2 && i < m->hp.layersCount) {
        llama_layer& l = m->layers[i];
        if (!strcmp(rest, "attention_norm.weight")) return l.attentionNorm;
        if (!strcmp(rest, "attention.wq.weight")) return l.wq;
        if (!strcmp(rest, "attention.wk.weight")) return l.wk;
        if (!strcmp(rest, "attention.wv.weight")) return l.wv;
        if (!strcmp(rest, "attention.wo.weight")) return l.wo;
        if (!strcmp(rest, "ffn_norm.weight")) return l.ffn_norm;
        if (!strcmp(rest, "feed_forward.w1.weight")) return l.w1;
        if (!strcmp(rest, "feed_forward.w2.weight")) return l.w2;
        if (!strcmp(rest, "feed_forward.w3.weight")) return l.w3;
    }
    return nullptr;
}

llama_model* llama_NewSyntheticModel(const llama_hparams* hp, uint64_t seed, uint32_t layer0, uint32_t layer1) {
    llama_model* m = alloc_model(hp, layer0, layer1);
    if (!m) return nullptr;
    const uint32_t d = hp->embdSize, F = m->ffSize;
    const float a_d = (float)sqrt(3.0 / (double)d), a_f = (float)sqrt(3.0 / (double)F);
    int rc = 0;
    rc |= fill(m->tokEmbeddings, seed, TID_TOK, (float)sqrt(3.0), 0.0f);
    rc |= fill(m->norm, seed, TID_NORM, 0.1f, 1.0f);
    rc |= fill(m->output, seed, TID_OUT, a_d, 0.0f);
    for (uint32_t i = m->layer0; i < m->layer1; i++) {
        llama_layer& l = m->layers[i];
        const uint32_t t = TID_LAYER0 + i * TID_PER_LAYER;
        rc |= fill(l.attentionNorm, seed, t + 0, 0.1f, 1.0f);
        rc |= fill(l.wq, seed, t + 1, a_d, 0.0f);
        rc |= fill(l.wk, seed, t + 2, a_d, 0.0f);
        rc |= fill(l.wv, seed, t + 3, a_d, 0.0f);
        rc |= fill(l.wo, seed, t + 4, a_d, 0.0f);
        rc |= fill(l.ffn_norm, seed, t + 5, 0.1f, 1.0f);
        rc |= fill(l.w1, seed, t + 6, a_d, 0.0f);
        rc |= fill(l.w2, seed, t + 7, a_f, 0.0f);
        rc |= fill(l.w3, seed, t + 8, a_d, 0.0f);
    }
    if (rc || lh_ctx_sync(model_ctx())) { llama_FreeModel(m); return nullptr; }
    return m;
}

void llama_FreeModel(llama_model* m) {
    if (!m) return;
    {   // contexts still alive (their plans address these buffers): the last one to go releases the model
        std::lock_guard<std::mutex> lk(m->mu);
        if (m->users > 0) { m->release_pending = true; return; }
    }
    free_model_now(m);
}
}  // extern "C"
static void free_model_now(llama_model* m) {
    free_tensor(m->tokEmbeddings); free_tensor(m->norm); free_tensor(m->output);
    for (llama_layer& l : m->layers) {
        free_tensor(l.attentionNorm); free_tensor(l.wq); free_tensor(l.wk); free_tensor(l.wv); free_tensor(l.wo);
        free_tensor(l.ffn_norm); free_tensor(l.w1); free_tensor(l.w2); free_tensor(l.w3);
    }
    delete m;
}
extern "C" {
void llama_ModelHParams(const llama_model* m, llama_hparams* out) { *out = m->hp; }
uint32_t llama_ModelFFSize(const llama_model* m) { return m->ffSize; }

// ---- ggjt v1 loader (llama.go:712-976): every tensor goes file -> bounce buffer -> HBM, no Go-heap-style copy kept
static float f16_to_f32(uint16_t h) {  // x448/float16 Float32() (llama.go:1005-1013): exact widening
    const uint32_t sign = (uint32_t)(h & 0x8000) << 16, exp = (h >> 10) & 0x1F;
    uint32_t man = h & 0x3FF, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else { int e = -1; do { e++; man <<= 1; } while (!(man & 0x400)); bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FF) << 13); }
    } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
static uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000;
    const int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
    uint32_t man = x & 0x7FFFFF;
    if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00 | (man ? 0x200 : 0));
    if (exp >= 31) return (uint16_t)(sign | 0x7C00);
    if (exp <= 0) {
        if (exp < -10) return (uint16_t)sign;
        man |= 0x800000;
        const uint32_t shift = (uint32_t)(14 - exp);
        uint32_t hm = man >> shift;
        const uint32_t rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(sign | hm);
    }
    const uint32_t hm = man >> 13, rem = man & 0x1FFF;
    uint16_t out = (uint16_t)(sign | ((uint32_t)exp << 10) | hm);
    if (rem > 0x1000 || (rem == 0x1000 && (hm & 1))) out++;
    return out;
}
static uint32_t rd_u32(FILE* f) { unsigned char b[4]; if (fread(b, 1, 4, f) != 4) return 0; return (uint32_t)b[3] << 24 | (uint32_t)b[2] << 16 | (uint32_t)b[1] << 8 | b[0]; }
static const uint32_t LLAMA_FILE_MAGIC = 0x67676a74u;

llama_model* llama_LoadModel(const char* fileName, uint32_t ctxSize) {
    FILE* f = fopen(fileName, "rb");
    if (!f) return (llama_model*)halt("[ERROR] cannot open model file");
    if (rd_u32(f) != LLAMA_FILE_MAGIC) { fclose(f); return (llama_model*)halt("[ERROR] Invalid model file! Wrong MAGIC in header"); }  // llama.go:722-732
    if (rd_u32(f) != 1) { fclose(f); return (llama_model*)halt("[ERROR] Invalid model file! Unsupported version"); }                     // llama.go:734-739
    llama_hparams hp;
    memset(&hp, 0, sizeof hp);
    hp.ctxSize = ctxSize;
    hp.vocabSize = rd_u32(f); hp.embdSize = rd_u32(f); hp.multSize = rd_u32(f); hp.headsCount = rd_u32(f);
    hp.layersCount = rd_u32(f); hp.rotCount = rd_u32(f); hp.f16 = rd_u32(f);  // llama.go:743-749
    if (!hp.vocabSize || !hp.embdSize || !hp.headsCount || !hp.layersCount || !hp.multSize) { fclose(f); return (llama_model*)halt("[ERROR] Invalid model file! Bad hyper-parameters"); }
    for (uint32_t i = 0; i < hp.vocabSize; i++) { const uint32_t len = rd_u32(f); fseek(f, (long)len + 4, SEEK_CUR); }  // vocab llama.go:799-811 (not on the hot path)
    llama_model* m = alloc_model(&hp, 0, 0);
    if (!m) { fclose(f); return nullptr; }
    lh_ctx* h = model_ctx();
    std::vector<float> bounce;
    std::vector<uint16_t> half;
    for (;;) {  // llama.go:889-969
        const uint32_t dims = rd_u32(f);
        if (dims < 1 || dims > 2) break;
        const uint32_t nameLen = rd_u32(f), dtype = rd_u32(f);
        uint32_t ne[2] = {1, 1};
        for (uint32_t i = 0; i < dims; i++) ne[i] = rd_u32(f);
        (void)ne;
        char name[256];
        if (nameLen >= sizeof name || fread(name, 1, nameLen, f) != nameLen) { fclose(f); llama_FreeModel(m); return (llama_model*)halt("[ERROR] bad tensor name in model file"); }
        name[nameLen] = 0;
        ml_tensor* t = llama_ModelTensor(m, name);
        if (!t) { fclose(f); llama_FreeModel(m); return (llama_model*)halt("[ERROR] Unknown tensor in model file"); }
        long off = ftell(f);
        off = (off + 31) & ~31L;  // 32-byte alignment llama.go:926-933
        fseek(f, off, SEEK_SET);
        const uint64_t n = nelements(t);
        bounce.resize(n);
        if (dtype == ML_TYPE_F16) {  // widened to f32 at load, llama.go:938-941
            half.resize(n);
            if (fread(half.data(), 2, n, f) != n) { fclose(f); llama_FreeModel(m); return (llama_model*)halt("[ERROR] Failed to read FP16 tensor from model!"); }
            for (uint64_t i = 0; i < n; i++) bounce[i] = f16_to_f32(half[i]);
        } else if (dtype == ML_TYPE_F32) {
            if (fread(bounce.data(), 4, n, f) != n) { fclose(f); llama_FreeModel(m); return (llama_model*)halt("[ERROR] Failed to read BIG FP32 chunk from model!"); }
        } else { fclose(f); llama_FreeModel(m); return (llama_model*)halt("[ERROR] Tensor data type is not supported yet!"); }  // llama.go:956-959
        if (lh_buf_upload(h, t->buf, 0, bounce.data(), n)) { g_err = lh_last_error(h); fclose(f); llama_FreeModel(m); return nullptr; }
    }
    fclose(f);
    return m;
}

static void wr_u32(FILE* f, uint32_t v) { unsigned char b[4] = {(unsigned char)v, (unsigned char)(v >> 8), (unsigned char)(v >> 16), (unsigned char)(v >> 24)}; fwrite(b, 1, 4, f); }
static int wr_tensor(FILE* f, const char* name, ml_tensor* t, int ftype) {  // scripts/convert-pth-to-ggml.py:196-232
    const uint32_t dims = t->dims;
    const bool as16 = ftype == 1 && dims == 2;
    wr_u32(f, dims); wr_u32(f, (uint32_t)strlen(name)); wr_u32(f, as16 ? 1u : 0u);
    for (uint32_t i = 0; i < dims; i++) wr_u32(f, t->ne[i]);
    fwrite(name, 1, strlen(name), f);
    long off = ftell(f);
    while (off % 32) { fputc(0, f); off++; }
    const uint64_t n = nelements(t);
    std::vector<float> host(n);
    if (ml_TensorRead(nullptr, t, host.data(), n)) return 1;
    if (as16) { std::vector<uint16_t> hh(n); for (uint64_t i = 0; i < n; i++) hh[i] = f32_to_f16(host[i]); fwrite(hh.data(), 2, n, f); }
    else fwrite(host.data(), 4, n, f);
    return 0;
}
int llama_SaveModel(const llama_model* mc, const char* fileName, int ftype) {
    llama_model* m = (llama_model*)mc;
    if (m->layer0 != 0 || m->layer1 != m->hp.layersCount) return halt_rc("llama_SaveModel: partial (layer-sharded) model");
    FILE* f = fopen(fileName, "wb");
    if (!f) return halt_rc("llama_SaveModel: cannot open file");
    wr_u32(f, LLAMA_FILE_MAGIC); wr_u32(f, 1);
    wr_u32(f, m->hp.vocabSize); wr_u32(f, m->hp.embdSize); wr_u32(f, m->hp.multSize); wr_u32(f, m->hp.headsCount);
    wr_u32(f, m->hp.layersCount); wr_u32(f, m->hp.embdSize / m->hp.headsCount); wr_u32(f, (uint32_t)ftype);
    for (uint32_t i = 0; i < m->hp.vocabSize; i++) { char tok[16]; const int len = snprintf(tok, sizeof tok, "<%u>", i); wr_u32(f, (uint32_t)len); fwrite(tok, 1, (size_t)len, f); const float s = -(float)i; fwrite(&s, 4, 1, f); }
    int rc = 0;
    rc |= wr_tensor(f, "tok_embeddings.weight", m->tokEmbeddings, ftype);
    rc |= wr_tensor(f, "norm.weight", m->norm, ftype);
    rc |= wr_tensor(f, "output.weight", m->output, ftype);
    static const char* names[9] = {"attention_norm.weight", "attention.wq.weight", "attention.wk.weight", "attention.wv.weight", "attention.wo.weight",
                                   "ffn_norm.weight", "feed_forward.w1.weight", "feed_forward.w2.weight", "feed_forward.w3.weight"};
    for (uint32_t i = 0; i < m->hp.layersCount && !rc; i++) {
        llama_layer& l = m->layers[i];
        ml_tensor* ts[9] = {l.attentionNorm, l.wq, l.wk, l.wv, l.wo, l.ffn_norm, l.w1, l.w2, l.w3};
        for (int k = 0; k < 9 && !rc; k++) { char nm[96]; snprintf(nm, sizeof nm, "layers.%u.%s", i, names[k]); rc |= wr_tensor(f, nm, ts[k], ftype); }
    }
    fclose(f);
    return rc;
}

// ---- context + Eval -----------------------------------------------------------------------------------
llama_context* llama_NewContext(llama_model* m, uint32_t ctxSize, int maxThreads, int useAVX, int useNEON) {  // llama.go:91-103
    g_err.clear();
    const uint64_t nlayers = m->layer1 - m->layer0;
    const uint64_t size = (uint64_t)m->hp.embdSize * nlayers * ctxSize;
    if (size > 0xFFFFFFFFull) return (llama_context*)halt("[HALT] KV cache exceeds uint32 element count (ml.Tensor.NE is uint32)");
    ml_context* mc = ml_NewContext(maxThreads, useAVX, useNEON);
    if (!mc) return nullptr;
    llama_context* c = new llama_context();
    c->model = m;
    c->ctxSize = ctxSize;
    m->hp.ctxSize = ctxSize;
    c->mlctx = mc;
    c->K = new_weight(1, (uint32_t)size, 1);  // zero-filled like Go's make([]float32)
    c->V = new_weight(1, (uint32_t)size, 1);
    if (!c->K || !c->V) { llama_ReleaseContext(c); return nullptr; }
    c->logits.assign(m->hp.vocabSize, 0.f);
    model_acquire(m);
    c->holds_model = true;
    return c;
}
void llama_ReleaseContext(llama_context* c) {  // llama.go:105-113
    if (!c) return;
    if (c->resident) lh_llama_destroy(c->resident);
    if (c->mlctx) lh_ctx_sync(c->mlctx->hip);
    eval_cache_free(c->decode_graph);
    free_tensor(c->K); free_tensor(c->V);   // per-pod KV caches go with their context (device memory returns to the pool)
    ml_ReleaseContext(c->mlctx);
    if (c->holds_model) model_release(c->model);
    delete c;
}
const float* llama_Logits(const llama_context* c) { return c->logits.data(); }
ml_context* llama_MLContext(llama_context* c) { return c->mlctx; }

}  // extern "C"
// The graph of llama.Eval (llama.go:232-387) over `model`, the KV tensors kK / kV and N token ids at position pastCount: expanded into
// `graph`, returns the logits node (NULL on a "[HALT]" condition).  One builder serves llama_Eval and llamago_DescribeEvalGraph.
static ml_tensor* build_eval_graph(ml_context* ctx0, llama_model* model, ml_tensor* kK, ml_tensor* kV, uint32_t ctxSize, const uint32_t* tokens, uint32_t N,
                                   uint32_t pastCount, ml_graph* graph, ml_tensor** embd_out = nullptr) {
    const uint32_t embdSize = model->hp.embdSize, layersCount = model->hp.layersCount;
    const uint32_t headsCount = model->hp.headsCount, rotCount = embdSize / headsCount;
    ml_tensor* embd = new_tensor(ML_TYPE_F32, 1, N, 1, 1, 1, nullptr, 0);  // :239-242 token ids as fp32
    if (embd_out) *embd_out = embd;
    for (uint32_t i = 0; i < N; i++) embd->data[i] = (float)tokens[i];
    ml_tensor* inpL = ml_GetRows(ctx0, model->tokEmbeddings, embd);         // :244
    for (uint32_t il = 0; il < layersCount; il++) {
        llama_layer& L = model->layers[il];
        ml_tensor* inpSA = inpL;
        ml_tensor* cur = ml_RMSNorm(ctx0, inpL);                              // :255
        cur = ml_Mul(ctx0, ml_Repeat(ctx0, L.attentionNorm, cur), cur);       // :258-259
        ml_tensor* Qcur = ml_MulMat(ctx0, L.wq, cur);                         // :263-265
        ml_tensor* Kcur = ml_MulMat(ctx0, L.wk, cur);
        ml_tensor* Vcur = ml_MulMat(ctx0, L.wv, cur);
        {                                                                     // :268-279
            ml_tensor* k = ml_View1D(ctx0, kK, N * embdSize, embdSize * (il * ctxSize + pastCount));
            ml_tensor* v = ml_View1D(ctx0, kV, N * embdSize, embdSize * (il * ctxSize + pastCount));
            if (!k || !v) return nullptr;
            if (ml_BuildForwardExpand(graph, ml_Copy(ctx0, Kcur, k)) || ml_BuildForwardExpand(graph, ml_Copy(ctx0, Vcur, v))) return nullptr;
        }
        ml_tensor* Q = ml_Permute(ctx0,                                       // :281-288
            ml_Rope(ctx0, ml_Copy(ctx0, Qcur, new_tensor(ML_TYPE_F32, 3, embdSize / headsCount, headsCount, N, 1, nullptr, 0, false)), pastCount, rotCount, 0),
            0, 2, 1, 3);
        ml_tensor* K = ml_Permute(ctx0,                                       // :290-297
            ml_Rope(ctx0, ml_Reshape3D(ctx0, ml_View1D(ctx0, kK, (pastCount + N) * embdSize, il * ctxSize * embdSize),
                                       embdSize / headsCount, headsCount, pastCount + N), pastCount, rotCount, 1),
            0, 2, 1, 3);
        ml_tensor* KQ = ml_MulMat(ctx0, K, Q);                                // :300
        ml_tensor* sc = new_tensor(ML_TYPE_F32, 1, 1, 1, 1, 1, nullptr, 0);   // :303-307
        sc->data[0] = (float)(1.0 / sqrt((double)embdSize / (double)headsCount));
        ml_tensor* KQScaled = ml_Scale(ctx0, KQ, sc);
        ml_tensor* KQMasked = ml_DiagMaskInf(ctx0, KQScaled, pastCount);      // :310
        ml_tensor* KQSoftMax = ml_SoftMax(ctx0, KQMasked);                    // :313
        ml_tensor* VTrans = ml_Copy(ctx0,                                     // :315-322
            ml_Permute(ctx0, ml_Reshape3D(ctx0, ml_View1D(ctx0, kV, (pastCount + N) * embdSize, il * ctxSize * embdSize),
                                          embdSize / headsCount, headsCount, pastCount + N), 1, 2, 0, 3),
            new_tensor(ML_TYPE_F32, 3, pastCount + N, embdSize / headsCount, headsCount, 1, nullptr, 0, false));
        ml_tensor* KQV = ml_MulMat(ctx0, VTrans, KQSoftMax);                  // :325
        ml_tensor* KQVMerged = ml_Permute(ctx0, KQV, 0, 2, 1, 3);             // :328
        cur = ml_Copy(ctx0, KQVMerged, new_tensor(ML_TYPE_F32, 2, embdSize, N, 1, 1, nullptr, 0, false));  // :331-333
        cur = ml_MulMat(ctx0, L.wo, cur);                                     // :336
        ml_tensor* inpFF = ml_Add(ctx0, cur, inpSA);                          // :340
        cur = ml_RMSNorm(ctx0, inpFF);                                        // :346
        cur = ml_Mul(ctx0, ml_Repeat(ctx0, L.ffn_norm, cur), cur);            // :349-351
        ml_tensor* tmp = ml_MulMat(ctx0, L.w3, cur);                          // :354
        cur = ml_MulMat(ctx0, L.w1, cur);                                     // :356
        cur = ml_Silu(ctx0, cur);                                             // :359
        cur = ml_Mul(ctx0, cur, tmp);                                         // :361
        cur = ml_MulMat(ctx0, L.w2, cur);                                     // :363
        cur = ml_Add(ctx0, cur, inpFF);                                       // :366
        inpL = cur;
    }
    inpL = ml_RMSNorm(ctx0, inpL);                                            // :374
    inpL = ml_Mul(ctx0, ml_Repeat(ctx0, model->norm, inpL), inpL);            // :377-379
    inpL = ml_MulMat(ctx0, model->output, inpL);                              // :384
    if (ml_BuildForwardExpand(graph, inpL)) return nullptr;                   // :387
    return inpL;
}
// ---- the graph of a one-token Eval, kept between calls -------------------------------------------------------------------------------
// llama.Eval builds its graph anew for every token (llama.go:232-387): ~1700 tensors, 53 us here, + 13 us to flatten them for the C side, + the
// release - a fortieth of what the GPU then needs for the token.  For a fixed token count the graph of Eval(N, past) has ONE structure, and
// every number in its flattened form (extents, strides, view offsets, the Rope / DiagMaskInf parameter values) is an affine function of `past`.
// So the first one-token Eval of a context builds the graph at three consecutive positions, takes the per-field differences of the flattened
// arrays, CHECKS on the third that they really are affine (a graph whose shape logic ever stops being so simply is not cached), keeps the first
// graph and array, and later calls write  value(past) = value(p0) + (past - p0) * difference  into the few hundred fields that move.  Nothing
// about the shapes is restated here: the builder above stays the only place that knows them.  tests/test_graph_twin.py compares the patched array
// with a fresh build field by field over many positions (llamago_DescribeEvalArray, no GPU needed).  LLAMAGO_NO_EVAL_CACHE=1 turns it off.
static std::atomic<int> g_keep_decode_graph{getenv("LLAMAGO_NO_EVAL_CACHE") ? 0 : 1};
struct eval_cache {
    llama_model* model = nullptr;
    uint32_t N = 0, p0 = 0, nl = 0, nn = 0;
    bool with_emb = false, failed = false;
    std::vector<ml_tensor*> owned;      // every tensor the kept build constructed (taken off the thread's garbage list)
    ml_graph* graph = nullptr;
    ml_tensor* embd = nullptr;          // the token-id leaf
    std::vector<lh_tensor> flat;        // the array at p0, patched in place
    struct Patch { uint32_t t, f; int64_t base, d; };   // f: 0-3 ne, 4-7 nb, 8 view_off
    std::vector<Patch> patches;
    struct HostPatch { float* p; float base, d; };      // parameter leafs whose VALUE moves with past (Rope, DiagMaskInf)
    std::vector<HostPatch> hpatches;
    uint32_t logits_index = 0;
    int emb_index = -1;
};
static void eval_cache_free(eval_cache* c) {
    if (!c) return;
    if (c->graph) delete c->graph;      // its tensors are in `owned`
    for (ml_tensor* t : c->owned) free_tensor(t);
    delete c;
}
static int64_t flat_field(const lh_tensor& t, uint32_t f) { return f < 4 ? (int64_t)t.ne[f] : f < 8 ? (int64_t)t.nb[f - 4] : (int64_t)t.view_off; }
static eval_cache* eval_cache_build(ml_context* ctx0, llama_model* model, ml_tensor* kK, ml_tensor* kV, uint32_t ctxSize, uint32_t N, uint32_t pastHint, bool with_emb) {
    eval_cache* c = new eval_cache();
    c->model = model; c->N = N; c->with_emb = with_emb; c->failed = true;
    if ((uint64_t)N + 2 > ctxSize) return c;
    c->p0 = std::min<uint32_t>(pastHint, ctxSize - N - 2);
    std::vector<lh_tensor> F[3];
    std::vector<std::vector<float>> H[3];           // host data of the owner leafs, in array order
    const std::vector<uint32_t> tokens(N, 0u);
    const bool save = g_gc_enabled;
    g_gc_enabled = true;
    bool ok = true;
    for (int b = 0; b < 3 && ok; b++) {
        ml_tensor* const mark = g_gc_head;
        ml_graph* g = ml_NewGraph();
        ml_tensor *embd = nullptr, *inpL = build_eval_graph(ctx0, model, kK, kV, ctxSize, tokens.data(), N, c->p0 + (uint32_t)b, g, &embd);
        ok = inpL != nullptr;
        uint32_t nl = 0, nn = 0;
        if (ok) {
            if (with_emb && inpL->src1) inpL->src1->want_output = true;
            flatten_graph(ctx0, g, &nl, &nn);
            F[b] = ctx0->flat;
            for (uint32_t i = 0; i < nl + nn; i++) {
                const lh_tensor& t = F[b][i];
                std::vector<float> h;
                if (t.host) h.assign(t.host, t.host + (uint64_t)t.ne[0] * t.ne[1] * t.ne[2] * t.ne[3]);
                H[b].push_back(std::move(h));
            }
        }
        if (b == 0 && ok) {
            c->graph = g; c->embd = embd; c->nl = nl; c->nn = nn;
            c->logits_index = inpL->last_index;
            c->emb_index = (with_emb && inpL->src1) ? (int)inpL->src1->last_index : -1;
            while (g_gc_head && g_gc_head != mark) { ml_tensor* t = g_gc_head; gc_unlink(t); c->owned.push_back(t); }
        } else {
            ml_FreeGraph(g);
            gc_free_down_to(mark);
        }
    }
    g_gc_enabled = save;
    if (!ok || F[1].size() != F[0].size() || F[2].size() != F[0].size()) return c;
    for (uint32_t i = 0; i < F[0].size(); i++) {
        const lh_tensor &a = F[0][i], &b1 = F[1][i], &b2 = F[2][i];
        for (const lh_tensor* o : {&b1, &b2})
            if (o->op != a.op || o->dtype != a.dtype || o->flags != a.flags || o->src0 != a.src0 || o->src1 != a.src1 || o->storage != a.storage || o->buf != a.buf ||
                (o->host == nullptr) != (a.host == nullptr)) return c;
        for (uint32_t f = 0; f < 9; f++) {
            const int64_t v0 = flat_field(a, f), d = flat_field(b1, f) - v0;
            if (flat_field(b2, f) - v0 != 2 * d) return c;
            if (d) c->patches.push_back({i, f, v0, d});
        }
        if (a.host && a.host != c->embd->data) {
            const std::vector<float>&h0 = H[0][i], &h1 = H[1][i], &h2 = H[2][i];
            if (h1.size() != h0.size() || h2.size() != h0.size()) return c;
            for (size_t k = 0; k < h0.size(); k++) {
                const float d = h1[k] - h0[k];
                if (h2[k] != h0[k] + 2.f * d) return c;
                if (d != 0.f) c->hpatches.push_back({const_cast<float*>(a.host) + k, h0[k], d});
            }
        }
    }
    c->flat = std::move(F[0]);
    c->failed = false;
    return c;
}
// the cached graph at position `past` with these tokens: c->flat is then what flatten_graph would have produced for a fresh build
static void eval_cache_patch(eval_cache* c, const uint32_t* tokens, uint32_t past) {
    for (uint32_t i = 0; i < c->N; i++) c->embd->data[i] = (float)tokens[i];
    const int64_t k = (int64_t)past - (int64_t)c->p0;
    for (const eval_cache::Patch& p : c->patches) {
        lh_tensor& t = c->flat[p.t];
        const int64_t v = p.base + p.d * k;
        if (p.f < 4) t.ne[p.f] = (uint32_t)v; else if (p.f < 8) t.nb[p.f - 4] = (uint64_t)v; else t.view_off = (uint64_t)v;
    }
    for (const eval_cache::HostPatch& h : c->hpatches) *h.p = h.base + h.d * (float)k;
}
extern "C" {

int llama_Eval(llama_context* lctx, llama_model* model, const uint32_t* tokens, uint32_t N, uint32_t pastCount) {  // llama.go:211-426
    if (model->layer0 != 0 || model->layer1 != model->hp.layersCount) return halt_rc("llama_Eval: layer-sharded model, use llamago_NewPipeline");
    const uint32_t ctxSize = lctx->ctxSize, vocabSize = model->hp.vocabSize;
    if (N == 0 || (uint64_t)pastCount + N > ctxSize) return halt_rc("llama_Eval: token window outside the context (the reference would index past the KV slice, llama.go:274)");
    ml_context* ctx0 = lctx->mlctx;
    static const bool timing = getenv("LLAMAGO_TIMING") != nullptr;  // stderr: host-side phases of one Eval in microseconds
    const auto tp0 = std::chrono::steady_clock::now();
    auto us_since = [](std::chrono::steady_clock::time_point a) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - a).count(); };
    static const bool no_fusion = getenv("LLAMAGO_NO_FUSION") && getenv("LLAMAGO_NO_FUSION")[0] == '1';
    if (N == 1 && !no_fusion && g_keep_decode_graph.load(std::memory_order_relaxed)) {   // the decode loop (server.go:153-217): the kept graph of a one-token Eval, moved to this position
        const bool with_emb = !lctx->embedding.empty();
        eval_cache* c = lctx->decode_graph;
        if (c && (c->model != model || c->with_emb != with_emb)) { eval_cache_free(c); c = lctx->decode_graph = nullptr; }
        if (!c) c = lctx->decode_graph = eval_cache_build(ctx0, model, lctx->K, lctx->V, ctxSize, 1, pastCount, with_emb);
        if (!c->failed) {
            g_err.clear();
            if (!ctx0 || !ctx0->hip) return halt_rc("ml_GraphCompute: no HIP context");
            eval_cache_patch(c, tokens, pastCount);
            const long t_build = us_since(tp0);
            const auto tp1 = std::chrono::steady_clock::now();
            ctx0->generation++;   // nodes of the caller's earlier graphs are no longer the last computed ones (ml_TensorRead)
            if (lh_graph_compute(ctx0->hip, c->flat.data(), c->nl, c->nn, LH_GRAPH_LAST_ROW_LOGITS)) return halt_rc(lh_last_error(ctx0->hip));
            const long t_compute = us_since(tp1);
            const auto tp2 = std::chrono::steady_clock::now();
            if (lh_node_read(ctx0->hip, c->logits_index, 0, lctx->logits.data(), vocabSize)) return halt_rc(lh_last_error(ctx0->hip));
            if (with_emb && c->emb_index >= 0 && lh_node_read(ctx0->hip, (uint32_t)c->emb_index, 0, lctx->embedding.data(), model->hp.embdSize)) return halt_rc(lh_last_error(ctx0->hip));
            if (timing) fprintf(stderr, "[llamago] Eval N=1: kept graph moved in %ld us (%zu + %zu fields), GraphCompute %ld us, logits read %ld us\n", t_build, c->patches.size(), c->hpatches.size(), t_compute, us_since(tp2));
            return 0;
        }
    }
    ml_graph* graph = ml_NewGraph();
    int rc = 1;
    const bool save = g_gc_enabled;
    g_gc_enabled = true;
    ml_tensor* const gc_mark = g_gc_head;
    do {
        ml_tensor* inpL = build_eval_graph(ctx0, model, lctx->K, lctx->V, ctxSize, tokens, N, pastCount, graph);
        if (!inpL) break;
        ml_tensor* embeddings = inpL->src1;   // llama.go:381: the lm_head's input, norm * weight rows
        if (!lctx->embedding.empty() && embeddings) embeddings->want_output = true;   // a fused plan then keeps it (LH_T_OUTPUT)
        const long t_build = us_since(tp0);
        const auto tp1 = std::chrono::steady_clock::now();
        {   // :389 — this caller reads only row N-1 of the result (:394-401) and says so
            const char* e = getenv("LLAMAGO_NO_FUSION");
            if (graph_compute(ctx0, graph, (e && e[0] == '1') ? LH_GRAPH_NO_FUSION : LH_GRAPH_LAST_ROW_LOGITS)) break;
        }
        // :394-401 — only the last token's logits are copied out
        const long t_compute = us_since(tp1);
        const auto tp2 = std::chrono::steady_clock::now();
        if (lh_node_read(ctx0->hip, inpL->last_index, (uint64_t)vocabSize * (N - 1), lctx->logits.data(), vocabSize)) { g_err = lh_last_error(ctx0->hip); break; }
        if (!lctx->embedding.empty()) {   // llama.go:414-419: row N-1 of `embeddings`
            if (lh_node_read(ctx0->hip, embeddings->last_index, (uint64_t)model->hp.embdSize * (N - 1), lctx->embedding.data(), model->hp.embdSize)) { g_err = lh_last_error(ctx0->hip); break; }
        }
        if (timing) fprintf(stderr, "[llamago] Eval N=%u: graph build %ld us, GraphCompute %ld us, logits read %ld us\n", N, t_build, t_compute, us_since(tp2));
        rc = 0;
    } while (0);
    ml_FreeGraph(graph);
    gc_free_down_to(gc_mark);  // whatever this Eval constructed and the graph did not reach (a halted build, an unused temporary) goes with it: nothing accumulates between the runtime.GC() points of llama.go:423
    g_gc_enabled = save;
    return rc;
}

// Harness extension (no counterpart in the reference; needs no GPU): the graph llama.Eval builds for a model of shape `hp`, as numbers.
// Per tensor 11 int32: op, ne[4], nb[4], src0, src1 — sources are indices into this same list (leafs first, then nodes, the order
// ml_GraphCompute walks), -1 = nil.  The checker library exports the same function over its own builders: tests compare the two
// lists so that the product's host mirror and the checker cannot drift apart unnoticed.  Returns the number of tensors.
}  // extern "C"
namespace {
struct shape_model {   // a model of shape `hp` whose tensors have extents and no storage: enough to BUILD graphs over (no GPU)
    llama_model m;
    std::vector<ml_tensor*> mine;
    ml_tensor *kK = nullptr, *kV = nullptr;
    ml_tensor* shape_leaf(uint32_t dims, uint32_t ne0, uint32_t ne1) { ml_tensor* t = new_leaf(ML_TYPE_F32, dims, ne0, ne1, 1, false); mine.push_back(t); return t; }
    shape_model(const llama_hparams* hp, uint32_t ctxSize) {
        m.hp = *hp;
        m.ffSize = ff_size(hp->embdSize, hp->multSize);
        m.layer0 = 0; m.layer1 = hp->layersCount;
        const uint32_t d = hp->embdSize, V = hp->vocabSize, F = m.ffSize;
        m.tokEmbeddings = shape_leaf(2, d, V); m.norm = shape_leaf(1, d, 1); m.output = shape_leaf(2, d, V);
        m.layers.assign(hp->layersCount, llama_layer{});
        for (llama_layer& l : m.layers) {
            l.attentionNorm = shape_leaf(1, d, 1); l.wq = shape_leaf(2, d, d); l.wk = shape_leaf(2, d, d); l.wv = shape_leaf(2, d, d); l.wo = shape_leaf(2, d, d);
            l.ffn_norm = shape_leaf(1, d, 1); l.w1 = shape_leaf(2, d, F); l.w2 = shape_leaf(2, F, d); l.w3 = shape_leaf(2, d, F);
        }
        const uint32_t kvn = d * hp->layersCount * ctxSize;
        kK = shape_leaf(1, kvn, 1); kV = shape_leaf(1, kvn, 1);
    }
    ~shape_model() {
        for (ml_tensor* t : mine) free_tensor(t);
        m.tokEmbeddings = m.norm = m.output = nullptr;
        m.layers.clear();
    }
};
}  // namespace
extern "C" {
int llamago_DescribeEvalGraph(const llama_hparams* hp, uint32_t ctxSize, uint32_t N, uint32_t pastCount, int32_t* out, uint32_t cap_tensors, uint32_t* n_leafs) {
    g_err.clear();
    if (!hp || !N || (uint64_t)pastCount + N > ctxSize) return -1;
    shape_model sm(hp, ctxSize);
    std::vector<uint32_t> tokens(N, 1u);
    ml_graph* g = ml_NewGraph();
    const bool save = g_gc_enabled;
    g_gc_enabled = true;
    ml_tensor* const gc_mark = g_gc_head;
    int count = -1;
    if (build_eval_graph(nullptr, &sm.m, sm.kK, sm.kV, ctxSize, tokens.data(), N, pastCount, g)) {
        std::unordered_map<const ml_tensor*, int> idx;
        std::vector<ml_tensor*> all(g->leafs);
        all.insert(all.end(), g->nodes.begin(), g->nodes.end());
        for (size_t i = 0; i < all.size(); ++i) idx[all[i]] = (int)i;
        count = (int)all.size();
        if (n_leafs) *n_leafs = (uint32_t)g->leafs.size();
        for (size_t i = 0; i < all.size() && i < cap_tensors && out; ++i) {
            const ml_tensor* t = all[i];
            int32_t* o = out + i * 11;
            o[0] = t->op;
            for (int k = 0; k < 4; ++k) { o[1 + k] = (int32_t)t->ne[k]; o[5 + k] = (int32_t)t->nb[k]; }
            o[9] = t->src0 ? idx.at(t->src0) : -1;
            o[10] = t->src1 ? idx.at(t->src1) : -1;
        }
    }
    ml_FreeGraph(g);
    gc_free_down_to(gc_mark);
    g_gc_enabled = save;
    return count;
}
// Harness extension (no GPU): the array llama_Eval hands to lh_graph_compute for N tokens (ids 1..N) at `pastQuery`, as numbers - per tensor 16
// int64: op, dtype, flags, ne[4], nb[4], src0, src1, storage, view_off, and the bit patterns of an owner leaf's host values folded into one
// number (0 without host data).  pastBuild < 0: from a fresh build at pastQuery (what every Eval did before round 6).  pastBuild >= 0: from
// the KEPT graph of the decode loop, learnt around pastBuild and moved to pastQuery (eval_cache) - tests require the two to be identical.
// Returns the number of tensors, -1 on bad arguments, -2 when the kept graph declined (its affine check failed).
void llamago_KeepDecodeGraph(int on) { g_keep_decode_graph.store(on ? 1 : 0); }
int llamago_DescribeEvalArray(const llama_hparams* hp, uint32_t ctxSize, uint32_t N, int64_t pastBuild, uint32_t pastQuery, int64_t* out, uint32_t cap_tensors, uint32_t* n_leafs) {
    g_err.clear();
    if (!hp || !N || (uint64_t)pastQuery + N > ctxSize || (pastBuild >= 0 && (uint64_t)pastBuild + N > ctxSize)) return -1;
    shape_model sm(hp, ctxSize);
    std::vector<uint32_t> tokens(N);
    for (uint32_t i = 0; i < N; i++) tokens[i] = i + 1;
    ml_context scratch;
    scratch.maxThreads = 1; scratch.hip = nullptr;
    const lh_tensor* T = nullptr;
    uint32_t nl = 0, nn = 0;
    eval_cache* c = nullptr;
    ml_graph* g = nullptr;
    const bool save = g_gc_enabled;
    ml_tensor* const gc_mark = g_gc_head;
    int count = -1;
    if (pastBuild >= 0) {
        c = eval_cache_build(&scratch, &sm.m, sm.kK, sm.kV, ctxSize, N, (uint32_t)pastBuild, false);
        if (c->failed) count = -2;
        else { eval_cache_patch(c, tokens.data(), pastQuery); T = c->flat.data(); nl = c->nl; nn = c->nn; }
    } else {
        g = ml_NewGraph();
        g_gc_enabled = true;
        if (build_eval_graph(&scratch, &sm.m, sm.kK, sm.kV, ctxSize, tokens.data(), N, pastQuery, g)) { flatten_graph(&scratch, g, &nl, &nn); T = scratch.flat.data(); }
    }
    if (T) {
        count = (int)(nl + nn);
        if (n_leafs) *n_leafs = nl;
        for (uint32_t i = 0; i < nl + nn && i < cap_tensors && out; ++i) {
            const lh_tensor& t = T[i];
            int64_t* o = out + (size_t)i * 16;
            o[0] = t.op; o[1] = t.dtype; o[2] = t.flags;
            for (int k = 0; k < 4; ++k) { o[3 + k] = t.ne[k]; o[7 + k] = (int64_t)t.nb[k]; }
            o[11] = t.src0; o[12] = t.src1; o[13] = t.storage; o[14] = (int64_t)t.view_off;
            uint64_t h = 0;
            if (t.host) {
                const uint64_t n = (uint64_t)t.ne[0] * t.ne[1] * t.ne[2] * t.ne[3];
                for (uint64_t k = 0; k < n; k++) { uint32_t bits; memcpy(&bits, t.host + k, 4); h = h * 1000003ull + bits + 1; }
            }
            o[15] = (int64_t)(h & 0x7fffffffffffffffull);
        }
    }
    if (g) ml_FreeGraph(g);
    gc_free_down_to(gc_mark);
    g_gc_enabled = save;
    eval_cache_free(c);
    return count;
}

static uint32_t argmax_f32(const float* x, uint32_t n) {  // strict >, lowest index wins ties
    // two passes: the maximum over eight independent lanes (vectorises; the one-pass `x[i] > x[best]` chain cost ~40 us on 32000 logits, a
    // tenth of what a token's whole host side may cost), then the first index that holds it.  NaNs as the reference's `x[i] > x[best]` loop treats
    // them: a NaN at index 0 wins (nothing compares greater than it), a NaN anywhere else is skipped - the lanes start at -inf and `x > m` is
    // false for a NaN, so it never enters a lane (seeding the lanes with x[0..7] let a NaN there blind its lane).
    if (n == 0 || x[0] != x[0]) return 0;
    float m[8];
    for (int j = 0; j < 8; j++) m[j] = -INFINITY;
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; j++) m[j] = x[i + j] > m[j] ? x[i + j] : m[j];
    float mx = m[0];
    for (int j = 1; j < 8; j++) mx = m[j] > mx ? m[j] : mx;
    for (; i < n; i++) mx = x[i] > mx ? x[i] : mx;
    for (uint32_t k = 0; k < n; k++) if (x[k] == mx) return k;
    return 0;   // (every element a NaN behind a first one that is not: cannot happen - x[0] then equals mx)
}
extern "C" uint32_t llamago_Argmax(const float* x, uint32_t n) { return argmax_f32(x, n); }
int llama_GreedyDecode(llama_context* lctx, llama_model* m, const uint32_t* prompt, uint32_t n_prompt, uint32_t n_predict, uint32_t* out_tokens,
                       float* step_logits) {  // loop shape of server.Do, server.go:153-217, one llama.Eval (= one ml_GraphCompute) per token
    uint32_t past = 0;
    const uint32_t V = m->hp.vocabSize, ringSize = lctx->ctxSize ? lctx->ctxSize : 1;
    std::vector<uint32_t> ring(ringSize, 0u), embd;   // lastNTokens: CtxSize zeros (server.go:127-138)
    uint64_t pos = 0;
    for (uint32_t i = 0; i < n_prompt; i++) ring[pos++ % ringSize] = prompt[i];   // server.go:193-197
    if (llama_Eval(lctx, m, prompt, n_prompt, past)) return 1;
    past += n_prompt;
    for (uint32_t s = 0; s < n_predict; s++) {
        const uint32_t id = argmax_f32(lctx->logits.data(), V);
        ring[pos++ % ringSize] = id;   // appendToken, server.go:207
        out_tokens[s] = id;
        if (step_logits) memcpy(step_logits + (uint64_t)s * V, lctx->logits.data(), (size_t)V * 4);
        if (s + 1 == n_predict) break;
        embd.assign(1, id);
        if (past + 1 > lctx->ctxSize) {   // context swap, server.go:160-172 (lastNTokens already holds id: it is re-fed and then evaluated once more)
            if (lctx->keepCount > past) { g_err = "llama_GreedyDecode: KeepCount beyond the context"; return 1; }
            const uint32_t leftCount = past - lctx->keepCount, n = leftCount / 2;
            past = lctx->keepCount;
            embd.clear();
            for (uint32_t i = 0; i < n; i++) embd.push_back(ring[(pos - n + i) % ringSize]);
            embd.push_back(id);
        }
        if (llama_Eval(lctx, m, embd.data(), (uint32_t)embd.size(), past)) return 1;
        past += (uint32_t)embd.size();
    }
    return 0;
}
int llamago_GreedyContinue(llama_context* lctx, llama_model* m, uint32_t token, uint32_t past, uint32_t n_steps, uint32_t* out_tokens) {
    if (!lctx || !m || !out_tokens) { g_err = "llamago_GreedyContinue: null argument"; return 1; }
    if ((uint64_t)past + n_steps > lctx->ctxSize) { g_err = "llamago_GreedyContinue: the steps run past the context window"; return 1; }
    const uint32_t V = m->hp.vocabSize;
    for (uint32_t s = 0; s < n_steps; s++) {
        if (llama_Eval(lctx, m, &token, 1, past + s)) return 1;
        token = argmax_f32(lctx->logits.data(), V);
        out_tokens[s] = token;
    }
    return 0;
}
// ModelParams.Embedding (llama.go:52): lctx.Embedding is allocated and every Eval leaves row N-1 of `embeddings` in it (llama.go:414-419)
void llamago_EnableEmbedding(llama_context* c) { if (c) c->embedding.assign(c->model->hp.embdSize, 0.f); }
const float* llama_Embedding(llama_context* c) { return (c && !c->embedding.empty()) ? c->embedding.data() : nullptr; }
void llamago_SetKeepCount(llama_context* c, uint32_t keep) {
    if (!c) return;
    c->keepCount = keep;
    if (c->resident) lh_llama_set_keep(c->resident, keep);
}

// ---- product extensions: device-resident decode loop, kernel timing, pipeline stage -----------------------
static lh_llama* make_stage(llama_model* m, lh_ctx* hip, lh_buf k_cache, lh_buf v_cache, uint32_t ctxSize) {
    std::vector<lh_llama_layer> ls(m->hp.layersCount);
    for (uint32_t i = m->layer0; i < m->layer1; i++) {
        llama_layer& l = m->layers[i];
        ls[i] = lh_llama_layer{l.attentionNorm->buf, l.wq->buf, l.wk->buf, l.wv->buf, l.wo->buf, l.ffn_norm->buf, l.w1->buf, l.w2->buf, l.w3->buf};
    }
    lh_llama_desc d;
    memset(&d, 0, sizeof d);
    d.vocab = m->hp.vocabSize; d.embd = m->hp.embdSize; d.heads = m->hp.headsCount; d.layers = m->hp.layersCount; d.ff = m->ffSize; d.ctx = ctxSize;
    d.layer0 = m->layer0; d.layer1 = m->layer1;
    d.tok_embeddings = m->tokEmbeddings ? m->tokEmbeddings->buf : 0;
    d.norm = m->norm ? m->norm->buf : 0;
    d.output = m->output ? m->output->buf : 0;
    d.layer = ls.data();
    d.k_cache = k_cache; d.v_cache = v_cache;
    d.weight_dtype = m->wtype;
    lh_llama* out = nullptr;
    if (lh_llama_create(hip, &d, &out)) { g_err = lh_last_error(hip); return nullptr; }
    return out;
}
static lh_llama* resident(llama_context* c) {
    if (!c->resident) {
        c->resident = make_stage(c->model, c->mlctx->hip, c->K->buf, c->V->buf, c->ctxSize);
        if (c->resident) lh_llama_set_keep(c->resident, c->keepCount);
    }
    return c->resident;
}
// ---- sampler (llama.go:455-707) on the device ----------------------------------------------------------------
int llamago_SampleDebug(ml_context* ctx, const float* logits, uint32_t logitsCount, const uint32_t* lastNTokens, uint32_t lastNTokensSize, uint32_t topK, float topP,
                        float temp, float repeatPenalty, uint64_t seed, uint64_t draw, uint32_t* token, uint32_t* cand_ids, float* cand_probs, uint32_t* n_keep) {
    if (!ctx) { g_err = "llama_SampleTopPTopK: no context"; return 1; }
    const lh_sample_params sp = {topK, topP, temp, repeatPenalty, seed};
    if (lh_sample_top_p_top_k(ctx->hip, logits, logitsCount, lastNTokens, lastNTokensSize, &sp, draw, token, cand_ids, cand_probs, n_keep)) {
        g_err = lh_last_error(ctx->hip);
        return 1;
    }
    return 0;
}
int llama_SampleTopPTopK(ml_context* ctx, const float* logits, uint32_t logitsCount, const uint32_t* lastNTokens, uint32_t lastNTokensSize, uint32_t topK, float topP,
                         float temp, float repeatPenalty, uint64_t seed, uint64_t draw, uint32_t* token) {
    return llamago_SampleDebug(ctx, logits, logitsCount, lastNTokens, lastNTokensSize, topK, topP, temp, repeatPenalty, seed, draw, token, nullptr, nullptr, nullptr);
}
int llama_SampleDecode(llama_context* c, llama_model* m, const uint32_t* prompt, uint32_t n_prompt, uint32_t n_predict, uint32_t topK, float topP, float temp,
                       float repeatPenalty, uint64_t seed, uint32_t* out_tokens) {  // server.go:127-217, resident on the device
    if (!c || c->model != m) { g_err = "llama_SampleDecode: context does not belong to this model"; return 1; }
    lh_llama* r = resident(c);
    if (!r) return 1;
    const lh_sample_params sp = {topK, topP, temp, repeatPenalty, seed};
    if (lh_llama_decode_sample(r, prompt, n_prompt, n_predict, c->ctxSize, &sp, out_tokens)) { g_err = lh_last_error(c->mlctx->hip); return 1; }
    return 0;
}

int llamago_DecodeGreedyResident(llama_context* c, uint32_t first_token, uint32_t past, uint32_t n_steps, uint32_t* out_tokens, float* logits_last) {
    lh_llama* r = resident(c);
    if (!r) return 1;
    if (lh_llama_decode_greedy(r, first_token, past, n_steps, out_tokens, logits_last)) return halt_rc(lh_last_error(c->mlctx->hip));
    return 0;
}
int llamago_ProfileDecode(llama_context* c, uint32_t token, uint32_t past, uint32_t repeats, lh_kernel_time* out, uint32_t cap) {
    lh_llama* r = resident(c);
    if (!r) return -1;
    const int n = lh_llama_profile_decode(r, token, past, repeats, out, cap);
    if (n < 0) g_err = lh_last_error(c->mlctx->hip);
    return n;
}
int llamago_Stage(llama_context* c, const uint32_t* tokens, const void* tokens_dev, const void* x_in_dev, void* x_out_dev, uint32_t n, uint32_t past,
                  void* logits_dev, void* argmax_dev) {
    lh_llama* r = resident(c);
    if (!r) return 1;
    if (lh_llama_stage(r, tokens, (const uint32_t*)tokens_dev, (const float*)x_in_dev, (float*)x_out_dev, n, past, (float*)logits_dev, (uint32_t*)argmax_dev))
        return halt_rc(lh_last_error(c->mlctx->hip));
    return 0;
}
// ---- the pods of one GPU in ONE weight pass (lh_batch): contexts of the same model, one KV cache each --------------------------
// All contexts must have been created on the same ml.Context stream: llama_NewContext makes one lh_ctx per context, so a batch is built
// over contexts that SHARE one - llamago_NewBatchContexts creates them.
struct llama_batch {
    ml_context* mlctx = nullptr;
    llama_model* model = nullptr;
    std::vector<ml_tensor*> K, V;
    std::vector<lh_llama*> pods;
    lh_batch* b = nullptr;
    uint32_t ctxSize = 0;
};
void llamago_FreeBatch(llama_batch* p) {
    if (!p) return;
    if (p->b) lh_batch_destroy(p->b);
    for (lh_llama* m : p->pods) lh_llama_destroy(m);
    if (p->mlctx) lh_ctx_sync(p->mlctx->hip);
    for (ml_tensor* t : p->K) free_tensor(t);
    for (ml_tensor* t : p->V) free_tensor(t);
    ml_ReleaseContext(p->mlctx);
    if (p->model) model_release(p->model);
    delete p;
}
// `pods` llama.Contexts (llama.go:91-103: one KV cache each) over one Model on one stream, bound into an lh_batch
llama_batch* llamago_NewBatch(llama_model* m, uint32_t ctxSize, uint32_t pods) {
    g_err.clear();
    if (!m || !pods) return (llama_batch*)halt("llamago_NewBatch: bad arguments");
    if (m->layer0 != 0 || m->layer1 != m->hp.layersCount) return (llama_batch*)halt("llamago_NewBatch: layer-sharded model, use llamago_NewPipeline");
    const uint64_t size = (uint64_t)m->hp.embdSize * m->hp.layersCount * ctxSize;
    if (size > 0xFFFFFFFFull) return (llama_batch*)halt("[HALT] KV cache exceeds uint32 element count (ml.Tensor.NE is uint32)");
    llama_batch* p = new llama_batch();
    p->ctxSize = ctxSize;
    p->mlctx = ml_NewContext(1, 0, 0);
    if (!p->mlctx) { delete p; return nullptr; }
    p->model = m;
    model_acquire(m);
    lh_ctx* hip = p->mlctx->hip;
    for (uint32_t i = 0; i < pods; i++) {
        ml_tensor* k = new_weight(1, (uint32_t)size, 1);
        ml_tensor* v = new_weight(1, (uint32_t)size, 1);
        if (k) p->K.push_back(k);
        if (v) p->V.push_back(v);
        if (!k || !v) { llamago_FreeBatch(p); return nullptr; }
        lh_llama* st = make_stage(m, hip, k->buf, v->buf, ctxSize);
        if (!st) { llamago_FreeBatch(p); return nullptr; }
        p->pods.push_back(st);
    }
    if (lh_batch_create(hip, p->pods.data(), pods, &p->b)) { g_err = lh_last_error(hip); llamago_FreeBatch(p); return nullptr; }
    return p;
}
int llamago_BatchBatched(llama_batch* p) { return p ? lh_batch_batched(p->b) : 0; }
// prompts -> for every pod its prompt Eval (server.go:185-192) and then n_predict - 1 decode steps of ALL pods per weight pass, greedy:
// out[i * n_predict + s] = s-th id of pod i (what llama_GreedyDecode returns for that prompt alone).  logits (optional): [pods][vocab]
// of the last tick.
int llamago_BatchGreedyDecode(llama_batch* p, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t n_predict, uint32_t* out, float* logits) {
    if (!p || !n_predict) return halt_rc("llamago_BatchGreedyDecode: bad arguments");
    lh_ctx* hip = p->mlctx->hip;
    const uint32_t B = (uint32_t)p->pods.size();
    if (lh_batch_set_sampler(p->b, nullptr, 0, nullptr, nullptr)) return halt_rc(lh_last_error(hip));
    if (lh_batch_prompt(p->b, prompts, n_prompt, nullptr, nullptr)) return halt_rc(lh_last_error(hip));
    std::vector<uint32_t> first(B);
    if (lh_batch_read_ids(p->b, first.data())) return halt_rc(lh_last_error(hip));
    for (uint32_t i = 0; i < B; i++) out[(size_t)i * n_predict] = first[i];
    if (n_predict == 1) return 0;
    std::vector<uint32_t> rest((size_t)B * (n_predict - 1));
    if (lh_batch_decode(p->b, first.data(), n_prompt, n_predict - 1, rest.data(), logits)) return halt_rc(lh_last_error(hip));
    for (uint32_t i = 0; i < B; i++)
        for (uint32_t s2 = 0; s2 + 1 < n_predict; s2++) out[(size_t)i * n_predict + 1 + s2] = rest[(size_t)i * (n_predict - 1) + s2];
    return 0;
}
void llamago_BatchSetKeepCount(llama_batch* p, uint32_t keep) {
    if (p) for (lh_llama* m : p->pods) lh_llama_set_keep(m, keep);
}
int llamago_BatchPrompt(llama_batch* p, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t* ids_out) {
    if (!p || !prompts || !n_prompt || !ids_out) return halt_rc("llamago_BatchPrompt: bad arguments");
    lh_ctx* hip = p->mlctx->hip;
    if (lh_batch_set_sampler(p->b, nullptr, 0, nullptr, nullptr)) return halt_rc(lh_last_error(hip));
    if (lh_batch_prompt(p->b, prompts, n_prompt, nullptr, nullptr)) return halt_rc(lh_last_error(hip));
    if (lh_batch_read_ids(p->b, ids_out)) return halt_rc(lh_last_error(hip));
    return 0;
}
int llamago_BatchSetSampler(llama_batch* p, uint32_t topK, float topP, float temp, float repeatPenalty, uint64_t seed, uint32_t ringSize) {
    if (!p) return halt_rc("llamago_BatchSetSampler: bad arguments");
    lh_ctx* hip = p->mlctx->hip;
    lh_sample_params sp = {topK, topP, temp, repeatPenalty, seed};
    if (lh_batch_set_sampler(p->b, &sp, ringSize, nullptr, nullptr)) return halt_rc(lh_last_error(hip));
    return 0;
}
int llamago_BatchTick(llama_batch* p, uint32_t* ids_out) {
    if (!p || !ids_out) return halt_rc("llamago_BatchTick: bad arguments");
    lh_ctx* hip = p->mlctx->hip;
    if (lh_batch_stage(p->b, nullptr, nullptr, nullptr, nullptr)) return halt_rc(lh_last_error(hip));
    if (lh_batch_read_ids(p->b, ids_out)) return halt_rc(lh_last_error(hip));
    return 0;
}

// ---- pods as pipeline streams over a layer-sharded model (server.go:84-106, 151) ----------------------
// One ml.Context (= one HIP stream) per rank carries every pod's stage and the RCCL p2p; each pod owns its KV cache like a
// llama.Context does (llama.go:91-98).  All scheduling happens below the C-ABI (lh_pipeline_run).
struct llama_pipeline {
    ml_context* mlctx = nullptr;
    llama_model* model = nullptr;
    lh_comm* comm = nullptr;
    lh_pipeline* pl = nullptr;
    std::vector<ml_tensor*> K, V;
    std::vector<lh_llama*> pods;
    bool holds_model = false;
};
int llamago_CommUniqueId(uint8_t* id) {
    lh_ctx* h = model_ctx();
    if (!h) return 1;
    if (lh_comm_unique_id(h, id)) return halt_rc(lh_last_error(h));
    return 0;
}
void llamago_FreePipeline(llama_pipeline* p) {
    if (!p) return;
    if (p->pl) lh_pipeline_destroy(p->pl);
    for (lh_llama* m : p->pods) lh_llama_destroy(m);
    if (p->comm) lh_comm_destroy(p->comm);
    if (p->mlctx) lh_ctx_sync(p->mlctx->hip);
    for (ml_tensor* t : p->K) free_tensor(t);
    for (ml_tensor* t : p->V) free_tensor(t);
    ml_ReleaseContext(p->mlctx);
    if (p->holds_model) model_release(p->model);
    delete p;
}
// id: the 128-byte RCCL unique id from rank 0 (llamago_CommUniqueId), or NULL with hooks (host-staged transport), or both NULL
// for an unsharded model (world must be 1).
// maxRows: rows (streams) one tick evaluates together in one pass over the weights (lh_pipeline_create_grouped); 0 = as many as fit.
llama_pipeline* llamago_NewPipelineGrouped(llama_model* m, uint32_t ctxSize, uint32_t pods, int rank, int world, const uint8_t* id, const lh_comm_hooks* hooks, uint32_t maxRows) {
    g_err.clear();
    if (!m || !pods || world < 1 || rank < 0 || rank >= world) return (llama_pipeline*)halt("llamago_NewPipeline: bad arguments");
    const uint64_t nlayers = m->layer1 - m->layer0, size = (uint64_t)m->hp.embdSize * nlayers * ctxSize;
    if (size > 0xFFFFFFFFull) return (llama_pipeline*)halt("[HALT] KV cache exceeds uint32 element count (ml.Tensor.NE is uint32)");
    llama_pipeline* p = new llama_pipeline();
    p->model = m;
    p->mlctx = ml_NewContext(1, 0, 0);
    if (!p->mlctx) { delete p; return nullptr; }
    model_acquire(m);
    p->holds_model = true;
    lh_ctx* hip = p->mlctx->hip;
    int rc = 0;
    if (id) rc = lh_comm_init(hip, rank, world, id, &p->comm);
    else if (hooks) rc = lh_comm_init_hooks(hip, rank, world, hooks, &p->comm);
    else if (world != 1) { llamago_FreePipeline(p); return (llama_pipeline*)halt("llamago_NewPipeline: a sharded model needs a communicator id or transport hooks"); }
    if (rc) { g_err = lh_last_error(hip); llamago_FreePipeline(p); return nullptr; }
    for (uint32_t i = 0; i < pods; i++) {
        ml_tensor* k = new_weight(1, (uint32_t)size, 1);
        ml_tensor* v = new_weight(1, (uint32_t)size, 1);
        if (k) p->K.push_back(k);
        if (v) p->V.push_back(v);
        if (!k || !v) { llamago_FreePipeline(p); return nullptr; }
        lh_llama* st = make_stage(m, hip, k->buf, v->buf, ctxSize);
        if (!st) { llamago_FreePipeline(p); return nullptr; }
        p->pods.push_back(st);
    }
    if (lh_pipeline_create_grouped(hip, p->comm, p->pods.data(), pods, maxRows, &p->pl)) { g_err = lh_last_error(hip); llamago_FreePipeline(p); return nullptr; }
    return p;
}
llama_pipeline* llamago_NewPipeline(llama_model* m, uint32_t ctxSize, uint32_t pods, int rank, int world, const uint8_t* id, const lh_comm_hooks* hooks) {
    return llamago_NewPipelineGrouped(m, ctxSize, pods, rank, world, id, hooks, 0);
}
uint32_t llamago_PipelineGroups(llama_pipeline* p) { return p ? lh_pipeline_groups(p->pl) : 0; }
// server.Do's loop with SampleTopPTopK after every Eval (server.go:201-204) for every stream of the pipeline; prompts on every rank
int llamago_PipelineRunSample(llama_pipeline* p, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps, uint32_t topK, float topP, float temp,
                              float repeatPenalty, uint64_t seed, uint32_t ringSize) {
    const lh_sample_params sp = {topK, topP, temp, repeatPenalty, seed};
    if (lh_pipeline_run_sample(p->pl, prompts, n_prompt, steps, &sp, ringSize)) return halt_rc(lh_last_error(p->mlctx->hip));
    return 0;
}
int llamago_PipelineRun(llama_pipeline* p, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps) {
    if (lh_pipeline_run(p->pl, prompts, n_prompt, steps)) return halt_rc(lh_last_error(p->mlctx->hip));
    return 0;
}
int llamago_PipelineSetKeepCount(llama_pipeline* p, uint32_t keep) { return p ? lh_pipeline_set_keep(p->pl, keep) : 1; }
int llamago_PipelineProfile(llama_pipeline* p, int on) { return p ? lh_pipeline_profile(p->pl, on) : 1; }
int llamago_PipelineStats(llama_pipeline* p, uint32_t* ticks, float* stage_ms, float* exchange_ms) {
    lh_pipeline_stats st;
    if (!p || lh_pipeline_stats_read(p->pl, &st)) return 1;
    if (ticks) *ticks = st.ticks;
    if (stage_ms) *stage_ms = st.stage_ms;
    if (exchange_ms) *exchange_ms = st.exchange_ms;
    return 0;
}
int llamago_PipelineHopProbe(llama_pipeline* p, uint32_t bytes, uint32_t iters, float* us_per_hop) {
    if (!p) return 1;
    if (lh_pipeline_hop_probe(p->pl, bytes, iters, us_per_hop)) return halt_rc(lh_last_error(p->mlctx->hip));
    return 0;
}
int llamago_PipelineTokens(llama_pipeline* p, uint32_t pod, uint32_t* out, uint32_t cap) {
    const int n = lh_pipeline_tokens(p->pl, pod, out, cap);
    if (n < 0) g_err = lh_last_error(p->mlctx->hip);
    return n;
}
int llamago_PipelineProfileDecode(llama_pipeline* p, uint32_t token, uint32_t past, uint32_t repeats, lh_kernel_time* out, uint32_t cap) {
    const int n = lh_llama_profile_decode(p->pods[0], token, past, repeats, out, cap);
    if (n < 0) g_err = lh_last_error(p->mlctx->hip);
    return n;
}
int llamago_PipelineSync(llama_pipeline* p) { return lh_ctx_sync(p->mlctx->hip) ? halt_rc(lh_last_error(p->mlctx->hip)) : 0; }

// Block-int8 (format in csrc/kernels_q8.h): quantise every weight MATRIX of the model in HBM
// (norm vectors and the embedding table, which is only gathered from, stay f32) and release the f32 copies.
int llamago_QuantizeModelQ8(llama_model* m) {
    if (m->wtype == ML_TYPE_Q8_0) return 0;
    { std::lock_guard<std::mutex> lk(m->mu);
      if (m->users > 0) return halt_rc("llamago_QuantizeModelQ8: contexts of this model are alive (their plans address the f32 weights); quantise before creating contexts"); }
    lh_ctx* h = model_ctx();
    if (!h) return 1;
    auto q = [&](ml_tensor* t) -> int {
        if (!t) return 0;
        lh_buf nb = 0;
        if (lh_buf_quantize_q8(h, t->buf, t->ne[1], t->ne[0], &nb)) return halt_rc(lh_last_error(h));
        lh_buf_free(h, t->buf);
        t->buf = nb;
        t->type = ML_TYPE_Q8_0;
        return 0;
    };
    int rc = q(m->output);
    for (uint32_t i = m->layer0; i < m->layer1 && !rc; i++) {
        llama_layer& l = m->layers[i];
        rc |= q(l.wq); rc |= q(l.wk); rc |= q(l.wv); rc |= q(l.wo); rc |= q(l.w1); rc |= q(l.w2); rc |= q(l.w3);
    }
    if (!rc) m->wtype = ML_TYPE_Q8_0;
    return rc;
}
int llamago_TimeComputes(llama_context* c, int on) { return c && c->mlctx && c->mlctx->hip ? lh_ctx_time_computes(c->mlctx->hip, on) : 1; }
int llamago_ComputeStats(llama_context* c, uint64_t* calls, double* wall_us, double* device_us) {
    lh_compute_stats st = {};
    if (!c || !c->mlctx || !c->mlctx->hip || lh_ctx_compute_stats(c->mlctx->hip, &st)) return 1;
    if (calls) *calls = st.calls;
    if (wall_us) *wall_us = st.wall_us;
    if (device_us) *device_us = st.device_us;
    return 0;
}
int llamago_Sync(llama_context* c) { return lh_ctx_sync(c->mlctx->hip) ? halt_rc(lh_last_error(c->mlctx->hip)) : 0; }

}  // extern "C"
