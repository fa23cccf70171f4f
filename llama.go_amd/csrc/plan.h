// csrc/plan.h — fused LLaMA plan: resolved model description + scratch + captured decode graph.
#pragma once
#include "common.h"
#include "kernels_common.h"
#include "attn_worklist.h"

namespace lh {

struct SampleState;

struct LayerW {
    const float *attn_norm = nullptr, *wq = nullptr, *wk = nullptr, *wv = nullptr, *wo = nullptr, *ffn_norm = nullptr, *w1 = nullptr, *w2 = nullptr, *w3 = nullptr;
    // block-int8 models: scale planes of the seven matrices (the pointers above are then the int8 planes)
    const float *s_wq = nullptr, *s_wk = nullptr, *s_wv = nullptr, *s_wo = nullptr, *s_w1 = nullptr, *s_w2 = nullptr, *s_w3 = nullptr;
    bool operator==(const LayerW& o) const {
        return attn_norm == o.attn_norm && wq == o.wq && wk == o.wk && wv == o.wv && wo == o.wo && ffn_norm == o.ffn_norm && w1 == o.w1 && w2 == o.w2 && w3 == o.w3;
    }
};

// Device-resolved description of llama.Model + one KV cache (llama.go:181-193, 173-178).
struct ModelDesc {
    uint32_t V = 0, d = 0, H = 0, hd = 0, L = 0, F = 0, ctx = 0;
    uint32_t layer0 = 0, layer1 = 0;  // layers evaluated by this plan
    uint32_t cache_layer0 = 0;        // layer stored in slot 0 of the caches
    const float *tok_emb = nullptr, *norm = nullptr, *output = nullptr;
    const float* s_output = nullptr;  // block-int8: scales of the output matrix
    int wtype = 0;                    // dtype of the weight matrices: 0 = f32, 7 = block-int8 (norms and embeddings stay f32)
    std::vector<LayerW> layers;       // indexed by absolute layer id
    float *kc = nullptr, *vc = nullptr;
    std::shared_ptr<std::vector<uint32_t>> kv_hist;   // the K cache buffer's token history (Buffer::kv_hist); not part of same()
    bool same(const ModelDesc& o) const {
        return V == o.V && d == o.d && H == o.H && hd == o.hd && L == o.L && F == o.F && ctx == o.ctx && layer0 == o.layer0 && layer1 == o.layer1 &&
               cache_layer0 == o.cache_layer0 && wtype == o.wtype && tok_emb == o.tok_emb && norm == o.norm && output == o.output && kc == o.kc && vc == o.vc && layers == o.layers;
    }
    bool first_stage() const { return layer0 == 0; }
    bool last_stage() const { return layer1 == L; }
};

struct Plan {
    lh_ctx* ctx = nullptr;
    ModelDesc md;
    // scratch, sized for n_cap rows
    uint32_t n_cap = 0;
    uint64_t scratch_gen = 0;                 // counts re-allocations of the scratch below (graphs captured elsewhere - lh_batch - compare it)
    float *xa = nullptr, *xb = nullptr, *h = nullptr, *qraw = nullptr, *kraw = nullptr, *vraw = nullptr, *q = nullptr, *attn = nullptr;
    float *a1 = nullptr, *a3 = nullptr, *g = nullptr, *logits = nullptr;
    uint16_t* s3 = nullptr;                   // block-int8 plans: the activations of the int8 matmuls as three bf16 planes each (kernels_stream_q8b.h):
                                              // [3][s3_rows][d] normalised rows, [3][s3_rows][d] merged attention heads, [3][s3_rows][F] gated rows
                                              // (fp32 plans: up to the 64 rows k_stream_b9 takes, kernels_stream_b9.h)
    uint32_t s3_rows = 0;
    float* emb = nullptr;                     // LH_T_OUTPUT on llama.Eval's `embeddings`: the final norm rows [emb_cap][d]
    uint32_t emb_cap = 0;
    float* attn_part = nullptr;               // split-T decode attention partials [H][chunks][hd + 2] (plans with ctx > 256)
    float *scores = nullptr, *vt = nullptr;   // large-N prefill attention: S[H][N][Tp], V^T[H][hd][Tp]
    uint64_t scores_cap = 0, vt_cap = 0;
    float* part = nullptr;                    // raw partial sums between the K-chunk launches of the short-prompt kernel [8][rows]
    uint64_t part_cap = 0;
    float* fa_part = nullptr;                 // prefill attention: partial (O, m, l) records of the query blocks that are cut by key range
    uint64_t fa_part_cap = 0;
    // the work list of the last (n, past) the prefill attention ran for: every layer of an Eval shares it
    FaWork fa_work;
    uint32_t* tokens_dev = nullptr;
    const double2* rope = nullptr;            // this plan's RoPE table (rotation width hd, >= ctx positions), resolved at plan_create
    // decode graph state
    StepParams* sp_dev = nullptr;
    StepParams* sp_host = nullptr;   // pinned
    uint32_t* out_tokens_dev = nullptr;
    uint32_t out_cap = 0;
    uint32_t* argmax_dev = nullptr;
    // captured decode graphs: one Eval(N=1) (embed .. logits); the same + argmax + advance (resident greedy loop) as 1 step and as
    // GRAPH_MULTI consecutive steps per launch; the same + device sampler + advance (resident sampling loop), 1 and GRAPH_MULTI steps.
    // Several steps per graph launch keep the GPU fed when the host is slow to submit (a loaded host measured 207 instead of 230 tok/s
    // with one launch per token) and shave the launch gap between tokens.
    enum { G_STEP = 0, G_ADV1, G_ADVN, G_SMP1, G_SMPN, G_COUNT };
    static constexpr uint32_t GRAPH_MULTI = 8;
    hipGraph_t graph[G_COUNT] = {};
    hipGraphExec_t exec[G_COUNT] = {};
    struct SampleState* ss_dev = nullptr;     // sampler parameters + counters (read by the captured sampler kernel)
    uint32_t* ring_dev = nullptr;             // lastNTokens ring
    uint32_t ring_cap = 0;
    uint32_t smp_topk = 0;                    // topK the sampler launches (and the captured graph) were chosen for
    uint32_t slot_counter = 0;   // round-robin over the pinned StepParams slots of eager (non-graph) steps
    bool use_graph = true;
    // Context swap of the generation loops (pkg/server/server.go:160-172): the token evaluated at every position of this plan's KV cache, as far as
    // the host knows it (first stage; HIST_UNKNOWN elsewhere), and ModelParams.KeepCount (llama.go:47).  Evals with host token ids record
    // themselves (through whichever plan over this cache they ran); the resident loops record what they produced when they synchronise.
    static constexpr uint32_t HIST_UNKNOWN = 0xFFFFFFFFu;
    std::shared_ptr<std::vector<uint32_t>> hist;   // shared by every plan over the same KV cache (Buffer::kv_hist)
    uint32_t keep = 0;
    void record(uint32_t pos, uint32_t tok) { if (hist && pos < hist->size()) (*hist)[pos] = tok; }
};

// The re-fed run of a context swap (server.go:166-171) for a stream whose window is full: positions [0, past) hold hist[], `pending` is the
// sampled token that has not been evaluated yet (the reference has already appended it to lastNTokens, server.go:207, so the run ENDS with it
// and it is then evaluated once more behind the run - restated as the reference does it).  Returns the n = (past - keep) / 2 tokens that are
// evaluated as ONE Eval at position keep; the stream continues with `pending` at position keep + n.  false: a token of the run is unknown.
bool swap_refeed_tokens(const Plan* p, uint32_t past, uint32_t pending, std::vector<uint32_t>* out);

// A batched Eval: n rows that belong to n DIFFERENT streams (the pods of a rank, server.go:88-101), evaluated in one pass over the weights.
struct BatchCtx {
    const BatchRow* rows;      // device: row -> (its stream's KV cache, its position)
    const uint32_t* tok_dev;   // device: the rows' token ids (first stage)
    float* attn_part;          // split-T attention partials [rows][H][chunks][hd + 2] (plans with ctx > 256), else nullptr
};

int plan_create(lh_ctx* ctx, const ModelDesc& md, Plan** out);
void plan_destroy(Plan* p);
Plan* plan_find_or_create(lh_ctx* ctx, const ModelDesc& md, int* rc);
int plan_ensure_rows(Plan* p, uint32_t n);
// One llama.Eval on the plan: tokens (host) or x_in (device) -> logits rows in p->logits ([n][V]) or x_out (non-last stage).
// bc != nullptr: the rows are those of a batched Eval (tokens_host / past unused; scratch must already hold n rows; logits for every row).
int plan_eval(Plan* p, const uint32_t* tokens_host, const float* x_in_dev, float* x_out_dev, uint32_t n, uint32_t past, bool last_row_only = false, const BatchCtx* bc = nullptr);
bool plan_batch_rows_ok(const Plan* p, uint32_t n);   // can n rows of different streams take one weight pass on this plan?
// enqueue the kernels of one decode step (N = 1), parameters from p->sp_dev
int plan_enqueue_decode(Plan* p, const float* x_in_dev, float* x_out_dev, bool with_argmax_advance, lh_kernel_time* prof, uint32_t prof_cap, uint32_t* prof_n);
int plan_decode_step(Plan* p, uint32_t token, uint32_t past);  // graph replay of one step; logits in p->logits
// `embeddings` of the last Eval (llama.go:381): RMSNorm * norm weight of the n final residual rows, into a plan-owned buffer [n][embd]
int plan_embeddings(Plan* p, uint32_t n, float** out);
void destroy_plans(lh_ctx* ctx);

// sample.hip
int sample_check(lh_ctx* ctx, const lh_sample_params* sp, uint32_t V);
int sample_launch(lh_ctx* ctx, const float* logits, uint32_t V, SampleState* st, uint32_t* ring, StepParams* sp, uint32_t* out_tokens, uint32_t* token_out,
                  uint32_t* dbg_ids, float* dbg_probs, uint32_t* dbg_keep, int advance, uint32_t topk_hint);

}  // namespace lh

struct lh_llama {
    lh_ctx* ctx;
    lh::Plan* plan;
};
