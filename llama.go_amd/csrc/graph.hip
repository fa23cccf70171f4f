// csrc/graph.hip — lh_graph_compute: the replacement of ml.GraphCompute (pkg/ml/ml.go:1411-1528).
//
// 1. resolve storage (persistent buffers, per-graph scratch arena, view offsets);
// 2. recognise the graph llama.Eval builds (pkg/llama/llama.go:211-389) by STRUCTURE (walking src0/src1 from
//    the final node) and run it as a fused plan (plan.hip);
// 3. otherwise — or with LH_GRAPH_NO_FUSION — run node by node, one kernel per op, exactly like the reference's
//    sequential walk (INIT/FINALIZE phases are no-ops for every implemented op, ml.go:1501-1526).
#include "plan.h"
#include <chrono>
#include "kernels_generic.h"
#include <math.h>
#include <algorithm>

namespace lh {

int gemm_small_n(lh_ctx* ctx, const float* w, const float* x, float* y, const float* resid, uint32_t M, uint32_t K, uint32_t n,
                 uint32_t ldx, uint32_t ldy, const char* name);  // plan.hip

enum { OP_NONE = 0, OP_ADD = 2, OP_MUL = 4, OP_REPEAT = 10, OP_SILU = 17, OP_RMS_NORM = 19, OP_MUL_MAT = 20, OP_SCALE = 21, OP_CPY = 22,
       OP_RESHAPE = 23, OP_VIEW = 24, OP_PERMUTE = 25, OP_TRANSPOSE = 26, OP_GET_ROWS = 27, OP_DIAG_MASK_INF = 28, OP_SOFT_MAX = 29, OP_ROPE = 30 };

static inline uint64_t nelem(const lh_tensor& t) { return (uint64_t)t.ne[0] * t.ne[1] * t.ne[2] * t.ne[3]; }
static inline bool contiguous(const lh_tensor& t) {  // ml.go:206-211
    return t.nb[0] == 4 && t.nb[1] == t.nb[0] * t.ne[0] && t.nb[2] == t.nb[1] * t.ne[1] && t.nb[3] == t.nb[2] * t.ne[2];
}
// furthest element touched + 1 (floats) through the tensor's strides
static inline uint64_t extent(const lh_tensor& t) {
    uint64_t e = 1;
    for (int i = 0; i < 4; ++i) e += (uint64_t)(t.ne[i] - 1) * (t.nb[i] / 4);
    return e;
}

// ---------------------------------------------------------------------------------------------------
// Structural matcher for the graph of llama.Eval
// ---------------------------------------------------------------------------------------------------
struct Matcher {
    lh_ctx* ctx;
    const lh_tensor* T;
    uint32_t total, n_leafs;
    ModelDesc md;
    uint32_t N = 0, past = 0;
    int kc_owner = -1, vc_owner = -1;
    int tokens_leaf = -1;
    int emb_node = -1;   // `embeddings` of llama.Eval (llama.go:381): the final norm * weight rows, source of the lm_head
    std::vector<uint32_t> tokens;

    // Every node the pattern walks over is marked; the match only holds if ALL nodes of the graph were claimed (extra outputs or
    // side-effect copies expanded into the same graph would otherwise be silently skipped by the fused plan).
    mutable std::vector<uint8_t> claimed;
    std::vector<uint32_t> cpy_nodes;
    bool is(int i, int op) const {
        if (i < 0 || (uint32_t)i >= total || T[i].op != op) return false;
        claimed[i] = 1;
        return true;
    }
    // shape + byte strides of a (permuted) view
    bool view_is(int i, uint32_t n0, uint32_t n1, uint32_t n2, uint64_t b0, uint64_t b1, uint64_t b2) const {
        const lh_tensor& t = T[i];
        return t.ne[0] == n0 && t.ne[1] == n1 && t.ne[2] == n2 && t.ne[3] == 1 && t.nb[0] == b0 && t.nb[1] == b1 && t.nb[2] == b2;
    }
    int s0(int i) const { return T[i].src0; }
    int s1(int i) const { return T[i].src1; }
    // persistent weight leaf with the expected shape -> device pointer.  Matrices may be f32 or block-int8 (all the same
    // dtype within a model: `want`); norm vectors are always f32.
    const float* weight(int i, uint32_t ne0, uint32_t ne1, const float** scales = nullptr, bool matrix = false) {
        if (i < 0 || (uint32_t)i >= n_leafs) return nullptr;
        const lh_tensor& t = T[i];
        if (t.op != OP_NONE || t.storage != i || !t.buf) return nullptr;
        if (t.ne[0] != ne0 || t.ne[1] != ne1 || t.ne[2] != 1 || t.ne[3] != 1) return nullptr;
        Buffer* b = find_buffer_fast(ctx, t.buf);
        if (!b || b->nfloats < (uint64_t)ne0 * ne1 || b->dtype != (int)t.dtype) return nullptr;
        if (!matrix) return b->dtype == 0 && contiguous(t) ? b->dev : nullptr;
        if (b->dtype != 0 && b->dtype != 7) return nullptr;
        if (b->dtype == 0 && !contiguous(t)) return nullptr;
        if (b->dtype == 7 && (b->rows != ne1 || b->cols != ne0)) return nullptr;
        if (wtype_seen < 0) wtype_seen = b->dtype;
        else if (wtype_seen != b->dtype) return nullptr;
        if (scales) *scales = b->scales;
        return b->dev;
    }
    int wtype_seen = -1;
    // h = Mul(gamma or Repeat(gamma, .), RMSNorm(x))  ->  returns x index, gamma pointer
    bool norm_mul(int h, int* x, const float** gamma) {
        if (!is(h, OP_MUL)) return false;
        int g = s0(h), r = s1(h);
        if (!is(r, OP_RMS_NORM)) return false;
        if (is(g, OP_REPEAT)) g = s0(g);
        *gamma = weight(g, md.d, 1);
        if (!*gamma) return false;
        *x = s0(r);
        return true;
    }
    const float* host_param(int i, uint32_t n) const {
        if (i < 0 || (uint32_t)i >= total) return nullptr;
        const lh_tensor& t = T[i];
        if (t.op != OP_NONE || !t.host || nelem(t) != n) return nullptr;
        return t.host;
    }
    // View1D(cache) -> Reshape3D -> [Rope] : returns the cache owner and the view offset
    bool cache_view3d(int r3, int* owner, uint64_t* off, uint32_t* ne2) const {
        if (!is(r3, OP_RESHAPE)) return false;
        const int v = s0(r3);
        if (!is(v, OP_VIEW)) return false;
        const lh_tensor& t = T[r3];
        if (t.ne[0] != md.hd || t.ne[1] != md.H) return false;
        *owner = T[v].storage;
        *off = T[v].view_off;
        *ne2 = t.ne[2];
        return T[*owner].buf != 0;
    }

    bool match_layer(int x_out, uint32_t il, int* x_in) {
        const uint32_t d = md.d, F = md.F, hd = md.hd, H = md.H;
        LayerW& L = md.layers[il];
        // x_out = Add(MulMat(w2, Mul(Silu(MulMat(w1,h2)), MulMat(w3,h2))), inpFF)      llama.go:354-366
        if (!is(x_out, OP_ADD)) return false;
        const int mm2 = s0(x_out), inpFF = s1(x_out);
        if (!is(mm2, OP_MUL_MAT)) return false;
        if (!(L.w2 = weight(s0(mm2), F, d, &L.s_w2, true))) return false;
        const int gm = s1(mm2);
        if (!is(gm, OP_MUL)) return false;
        const int sl = s0(gm), mm3 = s1(gm);
        if (!is(sl, OP_SILU) || !is(mm3, OP_MUL_MAT)) return false;
        const int mm1 = s0(sl);
        if (!is(mm1, OP_MUL_MAT)) return false;
        if (!(L.w1 = weight(s0(mm1), d, F, &L.s_w1, true)) || !(L.w3 = weight(s0(mm3), d, F, &L.s_w3, true))) return false;
        const int h2 = s1(mm1);
        if (s1(mm3) != h2) return false;
        int xff;
        if (!norm_mul(h2, &xff, &L.ffn_norm) || xff != inpFF) return false;
        // inpFF = Add(MulMat(wo, A), inpSA)                                             llama.go:336-340
        if (!is(inpFF, OP_ADD)) return false;
        const int mmo = s0(inpFF), inpSA = s1(inpFF);
        if (!is(mmo, OP_MUL_MAT) || !(L.wo = weight(s0(mmo), d, d, &L.s_wo, true))) return false;
        // A = Cpy(Permute(KQV), new2D)                                                  llama.go:328-333
        const int A = s1(mmo);
        if (!is(A, OP_CPY) || !is(s0(A), OP_PERMUTE)) return false;
        const int kqv = s0(s0(A));
        if (!is(kqv, OP_MUL_MAT)) return false;
        // KQV = MulMat(VTrans, S);  VTrans = Cpy(Permute(Reshape3D(View1D(V))), new3D) llama.go:315-325
        const int vt = s0(kqv), S = s1(kqv);
        if (!is(vt, OP_CPY) || !is(s0(vt), OP_PERMUTE)) return false;
        int vo; uint64_t voff; uint32_t vT;
        if (!cache_view3d(s0(s0(vt)), &vo, &voff, &vT)) return false;
        // S = SoftMax(DiagMaskInf(Scale(KQ, sc), past))                                 llama.go:303-313
        if (!is(S, OP_SOFT_MAX) || !is(s0(S), OP_DIAG_MASK_INF)) return false;
        const int dm = s0(S);
        if (!is(s0(dm), OP_SCALE)) return false;
        const int scn = s0(dm), kq = s0(scn);
        const float* pastp = host_param(s1(dm), 1);
        const float* scp = host_param(s1(scn), 1);
        if (!pastp || !scp || !is(kq, OP_MUL_MAT)) return false;
        if (*scp != (float)(1.0 / sqrt((double)d / (double)H))) return false;
        // KQ = MulMat(Permute(Rope(Reshape3D(View1D(K)), mode 1)), Permute(Rope(Cpy(MulMat(wq,h1), new3D), mode 0)))   llama.go:281-300
        const int Kp = s0(kq), Qp = s1(kq);
        if (!is(Kp, OP_PERMUTE) || !is(Qp, OP_PERMUTE) || !is(s0(Kp), OP_ROPE) || !is(s0(Qp), OP_ROPE)) return false;
        const int kr = s0(Kp), qr = s0(Qp);
        const float* kpar = host_param(s1(kr), 3);
        const float* qpar = host_param(s1(qr), 3);
        if (!kpar || !qpar) return false;
        int ko; uint64_t koff; uint32_t kT;
        if (!cache_view3d(s0(kr), &ko, &koff, &kT)) return false;
        const int qc = s0(qr);
        if (!is(qc, OP_CPY) || !is(s0(qc), OP_MUL_MAT)) return false;
        const int mmq = s0(qc);
        if (!(L.wq = weight(s0(mmq), d, d, &L.s_wq, true))) return false;
        const int h1 = s1(mmq);
        int xsa;
        if (!norm_mul(h1, &xsa, &L.attn_norm) || xsa != inpSA) return false;
        // shapes / parameters
        const lh_tensor& q3 = T[qc];
        const uint32_t n = q3.ne[2];
        if (q3.ne[0] != hd || q3.ne[1] != H) return false;
        const uint32_t p = (uint32_t)qpar[0];
        if ((uint32_t)kpar[0] != p || (uint32_t)*pastp != p) return false;
        if ((uint32_t)qpar[1] != hd || (uint32_t)kpar[1] != hd || (uint32_t)qpar[2] != 0 || (uint32_t)kpar[2] != 1) return false;
        if (kT != p + n || vT != p + n) return false;
        // the views must be the reference's permutations, not look-alikes: Q/K Permute(0,2,1,3) of [hd,H,*] (llama.go:288, 297),
        // V Permute(1,2,0,3) (llama.go:319), KQV [hd,N,H] merged back with Permute(0,2,1,3) (llama.go:328); all K-contiguous copies
        const uint64_t e4 = 4, dB = (uint64_t)d * 4, hB = (uint64_t)hd * 4;
        if (!view_is(Qp, hd, n, H, e4, dB, hB) || !view_is(Kp, hd, p + n, H, e4, dB, hB)) return false;
        if (!view_is(s0(vt), p + n, hd, H, dB, e4, hB)) return false;
        if (!view_is(kqv, hd, n, H, e4, hB, hB * n) || !view_is(s0(A), hd, H, n, e4, hB * n, hB)) return false;
        if (!view_is(kq, p + n, n, H, e4, (uint64_t)(p + n) * 4, (uint64_t)(p + n) * n * 4)) return false;
        if (il + 1 == md.L) { N = n; past = p; kc_owner = ko; vc_owner = vo; }
        else if (N != n || past != p || kc_owner != ko || vc_owner != vo) return false;
        if (koff != (uint64_t)il * md.ctx * d || voff != koff) return false;
        // K / V stores: Cpy(MulMat(wk|wv, h1), View1D(cache, d*(il*ctx + past)))         llama.go:274-278
        bool gotk = false, gotv = false;
        for (const uint32_t i : cpy_nodes) {   // (every Cpy node of the graph, in graph order: collected once - scanning all nodes here for every layer was
                                               // most of the 25 us a one-token match took)
            const int src = T[i].src0;
            if (!is(src, OP_MUL_MAT) || s1(src) != h1 || src == mmq) continue;
            const int dstv = T[i].src1;
            if (!is(dstv, OP_VIEW)) continue;
            const int own = T[dstv].storage;
            if (T[dstv].view_off != (uint64_t)d * ((uint64_t)il * md.ctx + p) || T[dstv].ne[0] != n * d) continue;
            const float* wsc = nullptr;
            const float* w = weight(s0(src), d, d, &wsc, true);
            if (!w) continue;
            if (own == ko && !gotk) { L.wk = w; L.s_wk = wsc; gotk = true; claimed[i] = 1; }
            else if (own == vo && !gotv) { L.wv = w; L.s_wv = wsc; gotv = true; claimed[i] = 1; }
        }
        if (!gotk || !gotv) return false;
        *x_in = inpSA;
        return true;
    }

    bool run() {
        claimed.assign(total, 0);
        if (!run_pattern()) return false;
        for (uint32_t i = n_leafs; i < total; ++i)
            if (!claimed[i]) return false;  // a node outside the Eval pattern: run the graph node by node
        // LH_T_OUTPUT: the host wants to read this node back (in the reference every Tensor.Data is readable after GraphCompute, ml.go:1411-1528).
        // A fused plan materialises the final node and `embeddings` (llama.go:381, read at llama.go:414-419); any other flagged intermediate
        // never exists in it, so such a graph runs node by node.
        for (uint32_t i = n_leafs; i + 1 < total; ++i)
            if ((T[i].flags & LH_T_OUTPUT) && (int)i != emb_node) return false;
        return true;
    }
    bool run_pattern() {
        const int fin = (int)total - 1;
        if (!is(fin, OP_MUL_MAT)) return false;
        const lh_tensor& ft = T[fin];
        const int wout = s0(fin);
        if (wout < 0 || (uint32_t)wout >= n_leafs) return false;
        md.d = T[wout].ne[0];
        md.V = T[wout].ne[1];
        if (!md.d || !md.V || ft.ne[0] != md.V) return false;
        if (!(md.output = weight(wout, md.d, md.V, &md.s_output, true))) return false;
        int x;
        if (!norm_mul(s1(fin), &x, &md.norm)) return false;
        emb_node = s1(fin);
        // count layers by walking the residual chain down to GetRows
        std::vector<int> outs;
        int cur = x;
        while (is(cur, OP_ADD)) {
            outs.push_back(cur);
            const int inpFF = s1(cur);
            if (!is(inpFF, OP_ADD)) return false;
            cur = s1(inpFF);
            if (outs.size() > 4096) return false;
        }
        if (!is(cur, OP_GET_ROWS) || outs.empty()) return false;
        md.L = (uint32_t)outs.size();
        md.layer0 = 0; md.layer1 = md.L; md.cache_layer0 = 0;
        md.layers.assign(md.L, LayerW());
        // head geometry from the last layer's Q copy: find it through the pattern of layer L-1
        {
            const int inpFF = s1(outs[0]);
            const int mmo = s0(inpFF);
            if (!is(mmo, OP_MUL_MAT)) return false;
            const int A = s1(mmo);
            if (!is(A, OP_CPY) || !is(s0(A), OP_PERMUTE)) return false;
            const int kqv = s0(s0(A));
            if (!is(kqv, OP_MUL_MAT)) return false;
            const lh_tensor& k = T[kqv];  // [hd, N, H]
            md.hd = k.ne[0];
            md.H = k.ne[2];
            if (!md.hd || md.hd * md.H != md.d) return false;
            const int mm2 = s0(outs[0]);
            if (!is(mm2, OP_MUL_MAT)) return false;
            md.F = T[s0(mm2)].ne[0];
        }
        // context size from the KV cache extent: embd*layers*ctx floats (llama.go:93)
        {
            const int inpFF = s1(outs[0]);
            const int kqv = s0(s0(s1(s0(inpFF))));
            const int vt = s0(kqv);
            if (!is(vt, OP_CPY) || !is(s0(vt), OP_PERMUTE) || !is(s0(s0(vt)), OP_RESHAPE) || !is(s0(s0(s0(vt))), OP_VIEW)) return false;
            const int own = T[s0(s0(s0(vt)))].storage;
            if (own < 0 || !T[own].buf) return false;
            Buffer* b = find_buffer_fast(ctx, T[own].buf);
            if (!b) return false;
            const uint64_t per = (uint64_t)md.d * md.L;
            if (b->nfloats % per) return false;
            md.ctx = (uint32_t)(b->nfloats / per);
        }
        cpy_nodes.clear();
        for (uint32_t i = n_leafs; i < total; ++i)
            if (T[i].op == OP_CPY) cpy_nodes.push_back(i);
        int xin = -1;
        for (uint32_t k = 0; k < md.L; ++k) {
            const uint32_t il = md.L - 1 - k;
            if (!match_layer(outs[k], il, &xin)) return false;
            if (il > 0 && xin != outs[k + 1]) return false;
        }
        // x_0 = GetRows(tok_embeddings, embd)                                         llama.go:239-244
        if (!is(xin, OP_GET_ROWS)) return false;
        if (!(md.tok_emb = weight(s0(xin), md.d, md.V))) return false;
        const float* ids = host_param(s1(xin), N);
        if (!ids) return false;
        tokens.resize(N);
        for (uint32_t i = 0; i < N; ++i) {
            tokens[i] = (uint32_t)ids[i];  // ids travel as fp32 (ml.go:1739)
            if (tokens[i] >= md.V) return false;
        }
        Buffer* kb = find_buffer_fast(ctx, T[kc_owner].buf);
        Buffer* vb = find_buffer_fast(ctx, T[vc_owner].buf);
        if (!kb || !vb || kb == vb || vb->nfloats != kb->nfloats) return false;
        md.kc = kb->dev;
        md.vc = vb->dev;
        if (!kb->kv_hist) kb->kv_hist = std::make_shared<std::vector<uint32_t>>();
        md.kv_hist = kb->kv_hist;   // the cache's token history (context swap of the generation loops): shared by every plan over this buffer
        md.wtype = wtype_seen < 0 ? 0 : wtype_seen;
        if ((uint64_t)past + N > md.ctx) return false;
        return true;
    }
};

// ---------------------------------------------------------------------------------------------------
// generic node-by-node execution
// ---------------------------------------------------------------------------------------------------
static TView view_of(const lh_tensor& t, float* p) {
    TView v;
    v.p = p;
    for (int i = 0; i < 4; ++i) { v.ne[i] = t.ne[i]; v.ns[i] = t.nb[i] / 4; }
    return v;
}
static inline dim3 grid_for(uint64_t total) { return dim3((unsigned)std::min<uint64_t>((total + 255) / 256, 65535)); }

static int run_node(lh_ctx* ctx, const lh_tensor* T, const std::vector<float*>& P, uint32_t i) {
    const lh_tensor& t = T[i];
    hipStream_t st = ctx->stream;
    auto V = [&](int k) { return view_of(T[k], P[k]); };
    switch (t.op) {
        case OP_NONE: case OP_RESHAPE: case OP_VIEW: case OP_PERMUTE: return 0;  // NOPs ml.go:1658-1663
        case OP_GET_ROWS: {
            const lh_tensor &a = T[t.src0], &b = T[t.src1];
            if (t.ne[0] != a.ne[0] || t.ne[1] != nelem(b) || a.nb[0] != 4) LH_FAIL(ctx, LH_ESHAPE, "[HALT]ComputeForwardGetRows : wrong dimensions!");
            if (b.host) {  // Go panics on src0.Data[r*NE[0]:] past the table (ml.go:1748); never let the GPU read out of range
                for (uint64_t k = 0; k < nelem(b); ++k)
                    if (!(b.host[k] >= 0.f) || (uint64_t)b.host[k] >= a.ne[1]) LH_FAIL(ctx, LH_EINVAL, "GetRows: row index %g outside the table of %u rows", (double)b.host[k], a.ne[1]);
            } else {
                LH_FAIL(ctx, LH_EUNSUPPORTED, "GetRows: indices must be a host leaf (they cannot be bounds-checked on the device)");
            }
            LH_LAUNCH(g_get_rows, dim3((unsigned)nelem(b)), dim3(256), 0, st, V(t.src0), V(t.src1), V(i));
            break;
        }
        case OP_RMS_NORM: {
            const lh_tensor& a = T[t.src0];
            LH_LAUNCH(g_rms_norm, dim3(a.ne[1] * a.ne[2] * a.ne[3]), dim3(256), 0, st, V(t.src0), V(i));
            break;
        }
        case OP_REPEAT: LH_LAUNCH(g_repeat, grid_for((uint64_t)t.ne[0] * t.ne[1]), dim3(256), 0, st, V(t.src0), V(i)); break;
        case OP_MUL: {
            const lh_tensor &a = T[t.src0], &b = T[t.src1];
            for (int k = 0; k < 4; ++k)
                if (a.ne[k] != b.ne[k] || a.ne[k] != t.ne[k]) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardMulFP32 : different shapes!");
            LH_LAUNCH(g_mul, grid_for(nelem(a)), dim3(256), 0, st, V(t.src0), V(t.src1), V(i));
            break;
        }
        case OP_ADD: {
            if (T[t.src1].nb[0] != 4) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardAddFP32 : [src1] is NOT contiguous!");
            LH_LAUNCH(g_add, grid_for(nelem(T[t.src0])), dim3(256), 0, st, V(t.src0), V(t.src1), V(i));
            break;
        }
        case OP_SILU: {
            if (!contiguous(T[t.src0])) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardSiluFP32 : [src0] is NOT contiguous!");
            if (!contiguous(t)) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardSiluFP32 : [dst] is NOT contiguous!");
            LH_LAUNCH(g_silu, grid_for(nelem(t)), dim3(256), 0, st, V(t.src0), V(i));
            break;
        }
        case OP_SCALE: {
            if (!contiguous(T[t.src0])) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardScaleFP32 : [src0] is NOT contiguous!");
            if (!contiguous(t)) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardScaleFP32 : [dst] is NOT contiguous!");
            LH_LAUNCH(g_scale, grid_for(nelem(t)), dim3(256), 0, st, V(i), (const float*)P[t.src1]);
            break;
        }
        case OP_CPY: {
            const lh_tensor& a = T[t.src0];
            if (!contiguous(t)) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardDupFP32 : [dst] is NOT contiguous!");
            if (nelem(t) != nelem(a)) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardDupFP32 : [dst] and [src0] capacities are different!");
            LH_LAUNCH(g_cpy, grid_for(nelem(a)), dim3(256), 0, st, V(t.src0), P[i], nelem(a));
            break;
        }
        case OP_DIAG_MASK_INF: LH_LAUNCH(g_diag_mask_inf, grid_for(nelem(t)), dim3(256), 0, st, V(i), (const float*)P[t.src1]); break;
        case OP_SOFT_MAX: {
            if (!contiguous(T[t.src0])) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardSoftMaxFP32 : [src0] is NOT contiguous!");
            if (!contiguous(t)) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardSoftMaxFP32 : [dst] is NOT contiguous!");
            LH_LAUNCH(g_soft_max, dim3(t.ne[1] * t.ne[2] * t.ne[3]), dim3(256), 0, st, V(i));
            break;
        }
        case OP_ROPE: {
            const lh_tensor& b = T[t.src1];
            if (nelem(b) != 3 || !b.host) LH_FAIL(ctx, LH_ESHAPE, "[HALT] ComputeForwardRopeFP32 : src1 has NOT EXACT 3 elements!");
            const uint32_t past = (uint32_t)b.host[0], dims = (uint32_t)b.host[1], mode = (uint32_t)b.host[2];
            if (dims == 0 || dims % 2 || dims > t.ne[0]) LH_FAIL(ctx, LH_ESHAPE, "Rope: dims %u not supported", dims);
            const uint32_t maxpos = (mode == 0 ? past : 0) + t.ne[2];
            const double2* table = nullptr;
            int rc = ensure_rope_table(ctx, maxpos + 1, dims, &table);
            if (rc) return rc;
            const uint64_t total = (uint64_t)t.ne[3] * t.ne[2] * t.ne[1] * (dims / 2);
            LH_LAUNCH(g_rope, grid_for(total), dim3(256), 0, st, V(i), table, past, dims, mode);
            break;
        }
        case OP_MUL_MAT: {
            const lh_tensor &a = T[t.src0], &b = T[t.src1];
            if (a.ne[0] != b.ne[0] || a.ne[2] != b.ne[2] || a.ne[3] != b.ne[3]) LH_FAIL(ctx, LH_ESHAPE, "MulMat: incompatible shapes (ml.go:290-292)");
            if (a.nb[0] != 4 || b.nb[0] != 4) LH_FAIL(ctx, LH_ESHAPE, "MulMat: operands must be contiguous along K (ml.go:1950, 1967)");
            if (a.dtype != 0) LH_FAIL(ctx, LH_EUNSUPPORTED, "MulMat: block-int8 weights are only supported inside the fused LLaMA plan");
            const bool plain2d = contiguous(a) && contiguous(b) && contiguous(t) && a.ne[2] == 1 && a.ne[3] == 1 && b.ne[2] == 1 && b.ne[3] == 1 &&
                                 a.ne[0] % 4 == 0 && a.ne[0] <= 24576 && a.ne[1] >= 256;
            if (plain2d) return gemm_small_n(ctx, P[t.src0], P[t.src1], P[i], nullptr, a.ne[1], a.ne[0], b.ne[1], a.ne[0], a.ne[1], "mul_mat");
            const uint64_t outs = (uint64_t)a.ne[1] * a.ne[2] * a.ne[3] * b.ne[1];
            LH_LAUNCH(g_mul_mat, dim3((unsigned)((outs + 3) / 4)), dim3(256), 0, st, V(t.src0), V(t.src1), V(i));
            break;
        }
        default: LH_FAIL(ctx, LH_EUNSUPPORTED, "[HALT] Please implement : op %d (the reference halts on it too, ml.go:1536-1700)", (int)t.op);
    }
    LH_HIP(ctx, hipGetLastError());
    return 0;
}

}  // namespace lh

using namespace lh;

extern "C" {

int lh_graph_compute(lh_ctx* ctx, const lh_tensor* T, uint32_t n_leafs, uint32_t n_nodes, uint32_t flags) {
    if (!ctx || !T) return LH_EINVAL;
    static const bool timing = getenv("LLAMAHIP_TIMING") != nullptr;   // stderr: host phases of a fused Eval in microseconds
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return (long)std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count() / 1000.0; };
    const auto tq0 = now();
    LH_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t total = n_leafs + n_nodes;
    if (total == 0) return LH_OK;
    // lh_ctx_time_computes: one event in front of everything this call puts on the stream, one behind it (mark_end, in front of the call's
    // last wait); the sums are taken when the call returns with both recorded
    struct ComputeTimer {
        lh_ctx* c; std::chrono::steady_clock::time_point t0;
        ~ComputeTimer() {
            if (!c->tc_on || !c->tc_end) return;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->tc_ev0, c->tc_ev1) != hipSuccess) return;
            c->tc_calls++; c->tc_dev_us += (double)ms * 1e3;
            c->tc_wall_us += (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count() / 1e3;
        }
    } compute_timer{ctx, tq0};
    ctx->tc_end = false;
    if (ctx->tc_on) LH_HIP(ctx, hipEventRecord(ctx->tc_ev0, ctx->stream));
    auto mark_end = [&]() { if (ctx->tc_on && hipEventRecord(ctx->tc_ev1, ctx->stream) == hipSuccess) ctx->tc_end = true; };
    // ---- validate indices
    for (uint32_t i = 0; i < total; ++i) {
        const lh_tensor& t = T[i];
        if (t.storage < 0 || (uint32_t)t.storage >= total) LH_FAIL(ctx, LH_EINVAL, "tensor %u: storage index %d out of range", i, t.storage);
        if (t.src0 >= (int)total || t.src1 >= (int)total) LH_FAIL(ctx, LH_EINVAL, "tensor %u: source index out of range", i);
        if (T[t.storage].storage != t.storage) LH_FAIL(ctx, LH_EINVAL, "tensor %u: storage owner %d is itself a view", i, t.storage);
        if (i >= n_leafs && (t.src0 >= (int)i || t.src1 >= (int)i)) {
            // nodes must come after their sources (post-order, ml.go:647-697) except leaf sources
            if ((t.src0 >= (int)i && (uint32_t)t.src0 >= n_leafs) || (t.src1 >= (int)i && (uint32_t)t.src1 >= n_leafs))
                LH_FAIL(ctx, LH_EINVAL, "node %u is listed before one of its sources", i);
        }
    }
    ctx->last_ptr.assign(total, nullptr);
    ctx->last_len.assign(total, 0);
    ctx->last_fused = 0;
    ctx->pre_index = -1;

    // ---- fused LLaMA plan?
    if (!(flags & LH_GRAPH_NO_FUSION)) {
        Matcher m;
        m.ctx = ctx; m.T = T; m.total = total; m.n_leafs = n_leafs;
        const auto tq1 = now();
        if (m.run()) {
            int rc = 0;
            const auto tq2 = now();
            Plan* p = plan_find_or_create(ctx, m.md, &rc);
            const auto tq3 = now();
            if (p) rc = plan_eval(p, m.tokens.data(), nullptr, nullptr, m.N, m.past, (flags & LH_GRAPH_LAST_ROW_LOGITS) != 0);
            const auto tq4 = now();
            if (!rc && m.emb_node >= 0 && (T[m.emb_node].flags & LH_T_OUTPUT)) {
                float* emb = nullptr;
                rc = plan_embeddings(p, m.N, &emb);   // the final norm * weight rows [N][embd] (the fused lm_head launches never write them out)
                if (!rc) { ctx->last_ptr[m.emb_node] = emb; ctx->last_len[m.emb_node] = (uint64_t)m.N * m.md.d; }
            }
            if (!rc && (flags & LH_GRAPH_LAST_ROW_LOGITS)) {
                // this caller reads row N - 1 of the final node and nothing else (llama.go:394-401): the row follows the kernels into pinned host
                // memory in front of the Eval's one synchronisation, and lh_node_read of that range is then a host copy (round 5: a second
                // synchronise + copy + synchronise cost ~20 us of the 4.5 ms a token takes through this route)
                const uint64_t V = m.md.V;
                if (ctx->out_pinned_floats < V) {
                    if (ctx->out_pinned) hipHostFree(ctx->out_pinned);
                    ctx->out_pinned = nullptr; ctx->out_pinned_floats = 0;
                    if (hipHostMalloc((void**)&ctx->out_pinned, V * 4, hipHostMallocDefault) == hipSuccess) ctx->out_pinned_floats = V;
                }
                if (ctx->out_pinned && hipMemcpyAsync(ctx->out_pinned, p->logits + (uint64_t)(m.N - 1) * V, V * 4, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess) {
                    ctx->pre_index = (int64_t)total - 1; ctx->pre_off = (uint64_t)(m.N - 1) * V; ctx->pre_n = V;
                }
            }
            if (!rc) {
                mark_end();
                LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (timing) fprintf(stderr, "[llamahip] graph_compute N=%u: validate %.1f us, match %.1f, find plan %.1f, enqueue %.1f, wait %.1f\n", m.N, us(tq0, tq1), us(tq1, tq2), us(tq2, tq3), us(tq3, tq4), us(tq4, now()));
                ctx->last_ptr[total - 1] = p->logits;  // [N][V], the final node's layout (ne0 = V, ne1 = N)
                ctx->last_len[total - 1] = (uint64_t)m.N * m.md.V;
                ctx->last_fused = 1;
                return LH_OK;
            }
            // Shapes outside what the fused kernels are built for (more rows per workgroup than threads, K beyond the register tile,
            // head widths the attention kernel cannot tile) are not an error of the caller: the reference's generic MulMat has no such
            // limits (ml.go:1976-2098), so the graph runs node by node instead.  It recomputes everything, cache writes included.
            if (rc != LH_EUNSUPPORTED && rc != LH_ESHAPE) return rc;
            LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
    }

    // ---- generic path: storage for every owner
    std::vector<uint64_t> need(total, 0);  // floats needed per owner
    for (uint32_t i = 0; i < total; ++i) {
        const lh_tensor& t = T[i];
        const uint64_t e = t.view_off + extent(t);
        need[t.storage] = std::max(need[t.storage], e);
    }
    std::vector<uint64_t> offs(total, 0);
    uint64_t arena = 0, stage = 0;
    for (uint32_t i = 0; i < total; ++i) {
        if ((uint32_t)T[i].storage != i) continue;
        if (T[i].buf) {
            Buffer* b = find_buffer(ctx->ds, T[i].buf);
            if (!b) LH_FAIL(ctx, LH_EINVAL, "tensor %u: unknown buffer %llu", i, (unsigned long long)T[i].buf);
            if (b->nfloats < need[i]) LH_FAIL(ctx, LH_ESHAPE, "tensor %u: views reach %llu floats, buffer holds %llu", i, (unsigned long long)need[i], (unsigned long long)b->nfloats);
            continue;
        }
        offs[i] = arena;
        arena += (need[i] * 4 + 255) & ~(uint64_t)255;
        if (T[i].host) stage += (nelem(T[i]) * 4 + 15) & ~(uint64_t)15;
    }
    int rc;
    if ((rc = ensure_arena(ctx, arena + 256))) return rc;
    if ((rc = ensure_staging(ctx, stage + 16))) return rc;
    std::vector<float*> P(total, nullptr);
    for (uint32_t i = 0; i < total; ++i) {
        if ((uint32_t)T[i].storage != i) continue;
        P[i] = T[i].buf ? find_buffer(ctx->ds, T[i].buf)->dev : (float*)(ctx->arena + offs[i]);
    }
    for (uint32_t i = 0; i < total; ++i) {
        if ((uint32_t)T[i].storage != i) P[i] = P[T[i].storage] + T[i].view_off;
        ctx->last_ptr[i] = P[i];
        ctx->last_len[i] = need[T[i].storage] - T[i].view_off;
    }
    // ---- upload host leafs (token ids, rope / mask / scale parameters, caller-filled inputs)
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));  // staging buffer reuse
    uint64_t so = 0;
    for (uint32_t i = 0; i < total; ++i) {
        if ((uint32_t)T[i].storage != i || T[i].buf || !T[i].host) continue;
        const uint64_t bytes = nelem(T[i]) * 4;
        memcpy(ctx->staging + so, T[i].host, bytes);
        LH_HIP(ctx, hipMemcpyAsync(P[i], ctx->staging + so, bytes, hipMemcpyHostToDevice, ctx->stream));
        so += (bytes + 15) & ~(uint64_t)15;
    }
    // ---- sequential walk (ml.go:1501-1526)
    for (uint32_t i = n_leafs; i < total; ++i)
        if ((rc = run_node(ctx, T, P, i))) return rc;
    mark_end();
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_node_read(lh_ctx* ctx, uint32_t index, uint64_t off, float* dst, uint64_t n) {
    if (!ctx || !dst) return LH_EINVAL;
    if (index >= ctx->last_ptr.size() || !ctx->last_ptr[index])
        LH_FAIL(ctx, LH_EINVAL, "lh_node_read: tensor %u was not materialised by the last graph (a fused plan keeps the final node and the nodes flagged LH_T_OUTPUT; flag it, or use LH_GRAPH_NO_FUSION)", index);
    if (off > ctx->last_len[index] || n > ctx->last_len[index] - off) LH_FAIL(ctx, LH_EINVAL, "lh_node_read: range outside tensor %u", index);
    if ((int64_t)index == ctx->pre_index && off == ctx->pre_off && n == ctx->pre_n) {   // already on the host (lh_graph_compute synchronised behind the copy)
        memcpy(dst, ctx->out_pinned, n * 4);
        return LH_OK;
    }
    LH_HIP(ctx, hipSetDevice(ctx->device));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LH_HIP(ctx, hipMemcpyAsync(dst, ctx->last_ptr[index] + off, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    LH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return LH_OK;
}

int lh_last_graph_fused(lh_ctx* ctx) { return ctx ? ctx->last_fused : 0; }

int lh_ctx_time_computes(lh_ctx* ctx, int on) {
    if (!ctx) return LH_EINVAL;
    LH_HIP(ctx, hipSetDevice(ctx->device));
    if (on && !ctx->tc_ev0) {
        LH_HIP(ctx, hipEventCreate(&ctx->tc_ev0));
        LH_HIP(ctx, hipEventCreate(&ctx->tc_ev1));
    }
    ctx->tc_on = on != 0; ctx->tc_end = false;
    ctx->tc_calls = 0; ctx->tc_wall_us = ctx->tc_dev_us = 0;
    return LH_OK;
}
int lh_ctx_compute_stats(lh_ctx* ctx, lh_compute_stats* out) {
    if (!ctx || !out) return LH_EINVAL;
    out->calls = ctx->tc_calls; out->wall_us = ctx->tc_wall_us; out->device_us = ctx->tc_dev_us;
    return LH_OK;
}

}  // extern "C"
