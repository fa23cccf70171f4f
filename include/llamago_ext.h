/* include/llamago_ext.h — the llamago_* exports of the host libraries: entry points with NO counterpart in the reference
 * (gotzmann/llama.go), needed by harnesses (tests/, bench.py), by the device-resident loops and by the multi-GPU / multi-pod
 * paths.  include/llamago.h stays the mirror of the reference's own names; everything else a host library exports is declared
 * HERE, and tests/test_abi.py diffs `nm -D` of both libraries against the two headers in both directions, so that the hand-written
 * ctypes / cgo bindings cannot drift from the C++ definitions unnoticed (the product host includes this header: signatures are
 * compiler-checked).
 *
 *   [both]     exported by llama.go_amd/lib/libllamago.so AND by the checker library oracle/liboracle.so
 *   [product]  libllamago.so only (needs the MI355X backend)
 */
#ifndef LLAMAGO_EXT_H
#define LLAMAGO_EXT_H
#include "llamago.h"
#include "llamahip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ---- [both] harness helpers ------------------------------------------------------------------------------------------- */
/* Releases every tensor the calling thread's constructors made and no graph freed: stands where the reference forces
 * runtime.GC() after each Eval (llama.go:423). */
void llamago_CollectGarbage(void);
/* The graph llama.Eval builds for a model of shape hp, as numbers (needs no GPU): per tensor 11 int32 = op, ne[4], nb[4], src0, src1
 * (indices into the same list, leafs first then nodes in ml.GraphCompute's order, -1 = nil).  Returns the tensor count. */
int llamago_DescribeEvalGraph(const llama_hparams* hp, uint32_t ctxSize, uint32_t N, uint32_t pastCount, int32_t* out, uint32_t cap_tensors, uint32_t* n_leafs);
/* The decode part of llama_GreedyDecode's loop on its own: from the context's present state (the caller has evaluated everything up to position
 * `past`), n_steps times { llama.Eval of ONE token = one ml_GraphCompute = one lh_graph_compute; the host argmax of its logits row }, starting with
 * `token`; out_tokens[s] = the id step s produced.  The contract route bench.py's headline times.  No context swap: past + n_steps
 * must stay inside the window. */
int llamago_GreedyContinue(llama_context* lctx, llama_model* m, uint32_t token, uint32_t past, uint32_t n_steps, uint32_t* out_tokens);
/* The greedy pick of the generation loops on host logits (strict >, the lowest index wins ties; a NaN at index 0 wins, NaNs elsewhere are skipped -
 * what `if x[i] > x[best] { best = i }` does).  Exported so that the product's vectorised form is tested against the checker's loop. */
uint32_t llamago_Argmax(const float* x, uint32_t n);
/* llama_SampleTopPTopK that also returns the kept candidates in rank order after the topP rescale (the reference's logitsID / probs
 * right before the random pick, llama.go:639-661). */
int llamago_SampleDebug(ml_context* ctx, const float* logits, uint32_t logitsCount, const uint32_t* lastNTokens, uint32_t lastNTokensSize, uint32_t topK, float topP,
                        float temp, float repeatPenalty, uint64_t seed, uint64_t draw, uint32_t* token, uint32_t* cand_ids, float* cand_probs, uint32_t* n_keep);
/* Block-int8 weight matrices (ML_TYPE_Q8_0, format ours: SURVEY §8a row 22).  Product: quantises every matrix in HBM and releases the
 * f32 copies; checker: replaces its weights by the dequantised values.  Before any context of the model exists. */
int llamago_QuantizeModelQ8(llama_model* m);

/* ---- [product] device plumbing ---------------------------------------------------------------------------------------- */
int llamago_DeviceCount(void);                       /* lh_device_count */
int llamago_HbmReadProbe(uint64_t bytes, uint32_t repeats, float* gbps);   /* lh_hbm_read_probe on the model context's device */
void llamago_SetStream(void* hip_stream);            /* HIP stream for contexts created from now on (NULL: a private one each) */
int llamago_Sync(llama_context* c);                  /* waits for the context's stream */
int llamago_TimeComputes(llama_context* c, int on);  /* lh_ctx_time_computes on the context llama_Eval computes in */
int llamago_ComputeStats(llama_context* c, uint64_t* calls, double* wall_us, double* device_us);   /* lh_ctx_compute_stats */
int llamago_LastGraphFused(ml_context* ctx);         /* 1 if the last ml_GraphCompute ran as the fused LLaMA plan */
int llamago_GraphComputeNoFusion(ml_context* ctx, ml_graph* g);   /* ml_GraphCompute node by node with the generic kernels (op-level parity) */

/* ---- [product] harness helper of the kept decode graph (no GPU needed) -------------------------------------------------- */
/* The array llama_Eval hands to lh_graph_compute for N tokens (ids 1..N) at pastQuery, as numbers: per tensor 16 int64 = op, dtype, flags, ne[4],
 * nb[4], src0, src1, storage, view_off, a fold of an owner leaf's host values (0 without).  pastBuild < 0: from a fresh build at pastQuery;
 * pastBuild >= 0: from the graph llama_Eval KEEPS between one-token calls, learnt around pastBuild and moved to pastQuery (host/llamago.cpp,
 * eval_cache) - tests require both to be identical.  Returns the tensor count, -1 bad arguments, -2 the kept graph declined. */
int llamago_DescribeEvalArray(const llama_hparams* hp, uint32_t ctxSize, uint32_t N, int64_t pastBuild, uint32_t pastQuery, int64_t* out, uint32_t cap_tensors, uint32_t* n_leafs);
/* 0: every llama_Eval builds its graph anew (what LLAMAGO_NO_EVAL_CACHE=1 sets at load); 1 (default): one-token Evals move the kept graph. */
void llamago_KeepDecodeGraph(int on);

/* ---- [product] device-resident loops on a llama.Context --------------------------------------------------------------- */
/* n_steps greedy decode steps without host round trips (lh_llama_decode_greedy). */
int llamago_DecodeGreedyResident(llama_context* c, uint32_t first_token, uint32_t past, uint32_t n_steps, uint32_t* out_tokens, float* logits_last);
/* HIP-event timing of every kernel class of one decode step (lh_llama_profile_decode); returns the class count. */
int llamago_ProfileDecode(llama_context* c, uint32_t token, uint32_t past, uint32_t repeats, lh_kernel_time* out, uint32_t cap);
/* One pipeline stage of Eval on the context's own KV cache (lh_llama_stage). */
int llamago_Stage(llama_context* c, const uint32_t* tokens, const void* tokens_dev, const void* x_in_dev, void* x_out_dev, uint32_t n, uint32_t past,
                  void* logits_dev, void* argmax_dev);

/* ModelParams.Embedding (llama.go:52, 88): from now on every llama_Eval also leaves row N-1 of `embeddings` (the final norm * weight rows,
 * llama.go:381, 414-419) in lctx.Embedding; llama_Embedding (llamago.h) returns it ([embd] floats; NULL when not enabled).  On the GPU the fused plan
 * never writes those rows out by itself: the Eval graph's node is flagged LH_T_OUTPUT. */
void llamago_EnableEmbedding(llama_context* c);
/* ModelParams.KeepCount (llama.go:47): the tokens a context swap keeps (server.go:166-167); 0 in the reference's own main.go.  The generation
 * loops (llama_GreedyDecode, llama_SampleDecode, llamago_DecodeGreedyResident) swap context at the window's end as server.Do does. */
void llamago_SetKeepCount(llama_context* c, uint32_t keep);

/* ---- [product] the pods of one GPU in ONE weight pass (lh_batch; server.go:88-101, 151) ------------------------------- */
typedef struct llama_batch llama_batch;
/* `pods` llama.Contexts (one KV cache each, llama.go:91-103) over one whole Model on one stream, bound into an lh_batch. */
llama_batch* llamago_NewBatch(llama_model* m, uint32_t ctxSize, uint32_t pods);
void llamago_FreeBatch(llama_batch* b);
int llamago_BatchBatched(llama_batch* b);            /* lh_batch_batched */
/* Every pod: its prompt as one Eval, then n_predict - 1 greedy steps of ALL pods per weight pass.  out[i * n_predict + s] = s-th id
 * of pod i (= llama_GreedyDecode of that prompt alone); logits (optional): [pods][vocab] of the last tick. */
int llamago_BatchGreedyDecode(llama_batch* b, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t n_predict, uint32_t* out, float* logits);
/* The twins of the Go shim's BatchHIP.Prompt / BatchHIP.Tick (go/ml_hip_pods.go): every pod's prompt as one Eval / one decode step of every pod
 * in one pass over the weights; ids_out[pods] = the ids produced.  A tick that would leave a pod's context window is an error
 * (llama.Eval's pastCount + N <= CtxSize), never a write past its KV cache. */
void llamago_BatchSetKeepCount(llama_batch* b, uint32_t keep);   /* ModelParams.KeepCount of every pod (see llamago_SetKeepCount) */
int llamago_BatchPrompt(llama_batch* b, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t* ids_out);
int llamago_BatchTick(llama_batch* b, uint32_t* ids_out);
/* lh_batch_set_sampler on the batch: the following ticks pick every pod's id with SampleTopPTopK (llama.go:455-707) instead of the argmax, every
 * pod seeded like a solo run; ringSize slots of lastNTokens per pod, empty.  May be called mid-stream (behind llamago_BatchPrompt and ticks). */
int llamago_BatchSetSampler(llama_batch* b, uint32_t topK, float topP, float temp, float repeatPenalty, uint64_t seed, uint32_t ringSize);

/* ---- [product] pods as pipeline streams over a layer-sharded model ---------------------------- */
typedef struct llama_pipeline llama_pipeline;
int llamago_CommUniqueId(uint8_t* id /* [LH_COMM_ID_BYTES] */);   /* lh_comm_unique_id on this process's device: rank 0 calls, any channel distributes */
/* This rank's stages of `pods` independent streams over the model's [layer0, layer1) (llama_NewSyntheticModel / loader with a layer
 * range), one KV cache per stream, one HIP stream per rank.  id = the 128 bytes from rank 0 (RCCL), or NULL with hooks (host-staged
 * transport), or both NULL for an unsharded model (world 1).  maxRows (Grouped): streams one tick evaluates together in one pass over
 * the rank's weights; 0 = as many as fit, 1 = every stream on its own. */
llama_pipeline* llamago_NewPipeline(llama_model* m, uint32_t ctxSize, uint32_t pods, int rank, int world, const uint8_t* id, const lh_comm_hooks* hooks);
llama_pipeline* llamago_NewPipelineGrouped(llama_model* m, uint32_t ctxSize, uint32_t pods, int rank, int world, const uint8_t* id, const lh_comm_hooks* hooks,
                                           uint32_t maxRows);
void llamago_FreePipeline(llama_pipeline* p);
uint32_t llamago_PipelineGroups(llama_pipeline* p);
int llamago_PipelineRun(llama_pipeline* p, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps);             /* lh_pipeline_run */
int llamago_PipelineRunSample(llama_pipeline* p, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps, uint32_t topK, float topP, float temp,
                              float repeatPenalty, uint64_t seed, uint32_t ringSize);                                            /* lh_pipeline_run_sample */
int llamago_PipelineSetKeepCount(llama_pipeline* p, uint32_t keep);                                               /* lh_pipeline_set_keep (same value on every rank) */
int llamago_PipelineProfile(llama_pipeline* p, int on);                                                          /* lh_pipeline_profile */
int llamago_PipelineStats(llama_pipeline* p, uint32_t* ticks, float* stage_ms, float* exchange_ms);             /* lh_pipeline_stats_read */
int llamago_PipelineHopProbe(llama_pipeline* p, uint32_t bytes, uint32_t iters, float* us_per_hop);             /* lh_pipeline_hop_probe */
int llamago_PipelineTokens(llama_pipeline* p, uint32_t pod, uint32_t* out, uint32_t cap);                                        /* lh_pipeline_tokens */
int llamago_PipelineProfileDecode(llama_pipeline* p, uint32_t token, uint32_t past, uint32_t repeats, lh_kernel_time* out, uint32_t cap);
int llamago_PipelineSync(llama_pipeline* p);

#ifdef __cplusplus
}
#endif
#endif
