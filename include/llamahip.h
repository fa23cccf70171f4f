/* include/llamahip.h — the drop-in C-ABI boundary of the MI355X backend (libllamahip.so).
 *
 * The reference (gotzmann/llama.go) has no plugin/FFI interface; the call that keeps its
 * ml.Tensor / ml.Graph operator surface intact and crosses Go -> C once per Eval is
 * ml.GraphCompute (pkg/ml/ml.go:1411-1528, called from pkg/llama/llama.go:389).  Every entry point
 * below is what a cgo shim inside package ml binds (INTEGRATION.md shows that shim); each cites the
 * reference interface it replaces.  Plain C: opaque handles, pointers and sizes, no torch/HIP types
 * (a HIP stream is passed as void*).
 *
 * Conventions
 *  - every function returns 0 on success or a negative LH_E* code and never aborts the process
 *    (the reference os.Exit(1)s on "[HALT]" conditions, e.g. ml.go:1538, 2116-2124; the shim maps a
 *    non-zero code back to that behaviour).  lh_last_error() returns the message.
 *  - thread-agnostic: a Go goroutine may migrate between OS threads between calls, so every entry
 *    point re-selects its device and uses the context's explicit stream.  One lh_ctx must not be
 *    used from two threads at once (the reference has one ml.Context per pod, server.go:151);
 *    registered weight buffers are shared read-only by all contexts of a device (server.go:45).
 *  - blocking: lh_graph_compute returns after the results are visible to lh_node_read, matching the
 *    wg.Wait() join of the reference (ml.go:1652).
 *  - no host pointer is retained past the call that received it (cgo pointer rule): weights are
 *    copied to HBM at registration.
 *  - ordering: an entry point that reads or writes HOST memory (uploads, reads, prompts, produced ids,
 *    logits) returns after the context's stream has drained; one that only takes DEVICE pointers
 *    (lh_llama_stage, lh_batch_stage, lh_comm_exchange over RCCL) enqueues on the context's stream and
 *    returns.  The private stream is non-blocking, i.e. NOT ordered with the null stream: the library
 *    itself waits for its stream before any synchronous (null-stream) copy, and a caller that touches
 *    the same buffers from another stream orders that stream against lh_ctx_stream() itself.
 */
#ifndef LLAMAHIP_H
#define LLAMAHIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LH_ABI_VERSION 1

enum { LH_OK = 0, LH_EINVAL = -1, LH_ENOMEM = -2, LH_EHIP = -3, LH_EUNSUPPORTED = -4, LH_ENODEVICE = -5, LH_ESHAPE = -6 };

typedef struct lh_ctx lh_ctx;   /* device side of one ml.Context (ml.go:50-57): stream, scratch arena, cached plans */
typedef uint64_t lh_buf;        /* handle of a persistent device buffer (weights, KV cache); 0 = none */

/* ---- contexts:  ml.NewContext ml.go:59-74 / (*Context).ReleaseContext ml.go:77-80 ------------- */
int lh_abi_version(void);
int lh_device_count(void);                      /* 0 when no GPU is visible (never an error) */
/* stream: an existing hipStream_t to enqueue on (e.g. the framework's current stream), or NULL for a
 * private non-blocking stream. */
int lh_ctx_create(int device, void* stream, lh_ctx** out);
void lh_ctx_destroy(lh_ctx* ctx);
const char* lh_last_error(lh_ctx* ctx);         /* ctx may be NULL: last error of the calling thread */
int lh_ctx_sync(lh_ctx* ctx);                   /* wait for everything enqueued on the context's stream */
/* Measurement aid: GB/s of a read-only stream over `bytes` of HBM (16-byte non-temporal loads, eight per lane in flight, one workgroup
 * per CU: the decode weight stream's access pattern with no arithmetic behind it) = what this box's memory delivers to a kernel that
 * does nothing else; the yardstick bench.py reports next to the nominal 8 TB/s (SURVEY 8d "vs measured achievable"). */
int lh_hbm_read_probe(lh_ctx* ctx, uint64_t bytes, uint32_t repeats, float* gbps);
void* lh_ctx_stream(lh_ctx* ctx);

/* ---- persistent tensors: weights resident after LoadModel (llama.go:975), KV cache of NewContext
 *      (llama.go:91-98).  `key` is a caller-chosen stable id (the Go shim uses the address of the
 *      tensor's backing array); registering an existing key returns the existing buffer.  key 0 =
 *      anonymous.  dtype is ml.DType (ml.go:85-94): 0 = f32; 7 = block-int8 (ours).  host may be NULL
 *      (buffer zero-filled, as Go's make([]float32) is). ---------------------------------------- */
int lh_tensor_register(lh_ctx* ctx, uint64_t key, int dtype, const uint32_t ne[4], int persistent,
                       const void* host_or_null, lh_buf* out);
int lh_buf_upload(lh_ctx* ctx, lh_buf buf, uint64_t off_floats, const float* host, uint64_t n);
int lh_buf_read(lh_ctx* ctx, lh_buf buf, uint64_t off_floats, float* dst, uint64_t n);
/* Fill with the synthetic-model generator: w[i] = offset + scale * u(seed, tensor_id, i), in HBM. */
int lh_buf_fill_synth(lh_ctx* ctx, lh_buf buf, uint64_t off_floats, uint64_t n, uint64_t seed, uint32_t tensor_id,
                      float scale, float offset);
/* Block-int8 (dtype 7; our format — the reference has none, ml.go:85-94, llama.go:956-959): quantise a registered fp32
 * matrix [rows][cols] on the device into a new buffer (d = max|w|/127 per 32 columns, q = rint(w/d)).  A dtype-7 buffer
 * can also be registered directly from 36-byte interchange blocks {float d; int8 q[32]} with lh_tensor_register. */
int lh_buf_quantize_q8(lh_ctx* ctx, lh_buf src_f32, uint32_t rows, uint32_t cols, lh_buf* out);
/* Dequantised fp32 values of a dtype-7 buffer (w = fl32(d*q)), for checks. */
int lh_buf_read_q8(lh_ctx* ctx, lh_buf buf, uint64_t off, float* dst, uint64_t n);
int lh_buf_free(lh_ctx* ctx, lh_buf buf);
void* lh_buf_devptr(lh_ctx* ctx, lh_buf buf);   /* raw device address (interop with a framework's collectives) */
uint64_t lh_buf_nfloats(lh_ctx* ctx, lh_buf buf);

/* ---- the graph:  ml.GraphCompute ml.go:1411-1528 ---------------------------------------------------
 * One flat array describes Graph.Leafs (first n_leafs entries) then Graph.Nodes in execution order
 * (ml.go:42-44).  lh_tensor mirrors ml.Tensor (ml.go:180-203); op and dtype use the reference's
 * numeric values (ml.go:133-174, 85-94).  Go slices alias invisibly to C, so view identity is carried
 * explicitly: `storage` is the index of the tensor that owns the bytes (itself if it owns them) and
 * `view_off` the offset in floats from that owner's first element. */
typedef struct lh_tensor {
    uint8_t op;          /* ml.optype */
    uint8_t dtype;       /* ml.DType */
    uint16_t flags;      /* LH_T_* */
    uint32_t ne[4];      /* Tensor.NE */
    uint64_t nb[4];      /* Tensor.NB, bytes, widened to 64-bit (the reference's uint32 strides cap at 4 GiB) */
    int32_t src0, src1;  /* indices into this array, -1 = nil */
    int32_t storage;     /* owner index (== own index when this tensor owns its bytes) */
    uint32_t reserved;
    uint64_t view_off;   /* floats from the owner's base */
    lh_buf buf;          /* owner only: registered persistent buffer, or 0 for per-graph scratch */
    const float* host;   /* owner leaf with buf == 0: host data uploaded for this call (token ids, op params); else NULL */
} lh_tensor;

enum { LH_T_OUTPUT = 1 /* the host wants to read this node back with lh_node_read (in the reference every Tensor.Data is readable after
                           GraphCompute).  Node-by-node execution materialises everything anyway.  A fused plan materialises the final node and,
                           when flagged, llama.Eval's `embeddings` (the final norm rows, llama.go:381, 414-419); a graph with any other flagged
                           intermediate is executed node by node instead */ };
enum {
    LH_GRAPH_NO_FUSION = 1,        /* run every node 1:1 with the generic kernels (debug / op-level parity tests) */
    LH_GRAPH_LAST_ROW_LOGITS = 2   /* caller reads only row N-1 of the final MulMat (what llama.Eval does: it builds the
                                      lm_head for all N rows, llama.go:384, and copies out the last, llama.go:394-401): a
                                      fused plan then evaluates the lm_head for that row only and leaves rows 0..N-2 of the
                                      final node unwritten.  Ignored by the node-by-node path. */
};

int lh_graph_compute(lh_ctx* ctx, const lh_tensor* tensors, uint32_t n_leafs, uint32_t n_nodes, uint32_t flags);
/* Read elements of a tensor of the LAST computed graph (flat, in storage order from the tensor's first
 * element).  Replaces the host's direct reads of Tensor.Data after GraphCompute (llama.go:394-401). */
int lh_node_read(lh_ctx* ctx, uint32_t index, uint64_t off_floats, float* dst, uint64_t n);
/* 1 if the last lh_graph_compute ran as a fused LLaMA plan, 0 if node-by-node. */
int lh_last_graph_fused(lh_ctx* ctx);
/* Timing of a context's lh_graph_compute calls, the measurement SURVEY 8d config 2 names ("hipEvent around each lh_graph_compute"):
 * device_us = sum over the calls of the time between an event recorded on the stream when the call starts and one recorded behind its last
 * launch / copy; wall_us = sum of the host time spent inside the calls (validate, match, enqueue, the wait, the pinned copy of the row).
 * Off by default (two event records per call); lh_ctx_time_computes(ctx, on) zeroes the sums. */
typedef struct lh_compute_stats {
    uint64_t calls;
    double wall_us;
    double device_us;
} lh_compute_stats;
int lh_ctx_time_computes(lh_ctx* ctx, int on);
int lh_ctx_compute_stats(lh_ctx* ctx, lh_compute_stats* out);

/* Route log (test instrumentation, process-wide): while it is on, every kernel launch of the library notes the launched kernel's family - the
 * __global__'s name without template arguments.  tests/test_gpu_zz_routes.py switches it on for the whole GPU suite and asserts at the end that
 * every __global__ of the product sources was reached.  lh_route_log(1) clears the set and starts, lh_route_log(0) stops; lh_route_names writes the
 * names, newline-separated and NUL-terminated, into buf (up to cap bytes) and returns the number of bytes the full list needs. */
int lh_route_log(int on);
int64_t lh_route_names(char* buf, uint64_t cap);

/* ---- convenience layer over the same fused plan executor (harnesses, bench, pipeline stages) ------
 * Describes the weights of llama.Model (llama.go:181-193) + one KV cache (llama.go:173-178) for the
 * layer range [layer0, layer1) held by this process. */
typedef struct lh_llama_layer { lh_buf attention_norm, wq, wk, wv, wo, ffn_norm, w1, w2, w3; } lh_llama_layer;
typedef struct lh_llama_desc {
    uint32_t vocab, embd, heads, layers, ff, ctx; /* whole-model hyper-parameters (llama.go:149-158, 761) */
    uint32_t layer0, layer1;                      /* this stage's layers; [0, layers) = whole model */
    lh_buf tok_embeddings, norm, output;          /* needed on the first / last stage only (0 elsewhere) */
    const lh_llama_layer* layer;                  /* `layers` entries, only [layer0, layer1) are read */
    lh_buf k_cache, v_cache;                      /* embd * (layer1-layer0) * ctx floats each; layer il at slot (il-layer0) */
    int weight_dtype;                             /* 0 = f32, 7 = block-int8 matrices */
} lh_llama_desc;
typedef struct lh_llama lh_llama;

int lh_llama_create(lh_ctx* ctx, const lh_llama_desc* desc, lh_llama** out);
void lh_llama_destroy(lh_llama* m);
/* llama.Eval (llama.go:211-426) on a whole model: logits of the last token to host (vocab floats). */
int lh_llama_eval(lh_llama* m, const uint32_t* tokens, uint32_t n, uint32_t past, float* logits_host);
/* ModelParams.KeepCount (llama.go:47) of this context: the first tokens a context swap keeps.  The generation loops below do what
 * server.Do does when the window is full (pkg/server/server.go:160-172: leftCount = pastCount - KeepCount; pastCount = KeepCount; the
 * last leftCount / 2 entries of lastNTokens - which already holds the pending token - are re-fed in front of it and evaluated as one
 * Eval): they go on generating instead of failing.  The swap needs the tokens of the window, which the context knows from the Evals
 * that went through it with host token ids and from its own resident loops; otherwise the loop fails with LH_EINVAL at the window's
 * end.  A single Eval past the window (lh_llama_eval, lh_llama_stage, lh_graph_compute) stays an error, as llama.Eval would panic. */
int lh_llama_set_keep(lh_llama* m, uint32_t keep);
/* Device-resident greedy decode: step i evaluates one token at position past+i; the argmax (lowest
 * index on ties) is taken on the GPU and feeds step i+1 without a host round trip.  out_tokens[i]
 * = id produced by step i.  logits_last_host (optional) receives the final step's logits.  Past the
 * window the loop swaps context (above): one host synchronisation per (ctx - keep) / 2 tokens. */
int lh_llama_decode_greedy(lh_llama* m, uint32_t first_token, uint32_t past, uint32_t n_steps,
                           uint32_t* out_tokens, float* logits_last_host);
/* One pipeline stage of Eval for a layer-sharded model (residual stream in/out in device memory,
 * n rows of embd floats).  First stage: x_in = NULL and token ids given either on the host (tokens) or,
 * for n = 1, in device memory (tokens_dev: the id the last stage produced, received over xGMI, without
 * a host round trip).  Last stage: writes logits of the last token to logits_dev (vocab floats) and its
 * argmax to *argmax_dev (both device pointers, optional).  Asynchronous on the context's stream. */
int lh_llama_stage(lh_llama* m, const uint32_t* tokens, const uint32_t* tokens_dev, const float* x_in_dev, float* x_out_dev,
                   uint32_t n, uint32_t past, float* logits_dev, uint32_t* argmax_dev);
/* ---- sampler on the device --------------------------------------------------------------
 * SampleTopPTopK (llama.go:455-707): repeat penalty over the whole lastNTokens ring (llama.go:497-525), sort + topK
 * (llama.go:548-567), softmax with f64 exp (llama.go:581-609), topP cut (llama.go:623-639), then the reference's
 * "probs^2 * rand^2, first maximum" pick (llama.go:661-673).  The reference seeds math/rand from the clock on every
 * call (llama.go:658); here the uniforms come from a counter-based generator over (seed, call index, rank), so a
 * run is reproducible.  Ties in the sort (unspecified in the reference, sort.Slice) go to the lower token id. */
typedef struct lh_sample_params {
    uint32_t top_k;        /* ModelParams.TopK, 40 (main.go:87); 1..min(vocab, 1024) */
    float top_p;           /* 0.95 (main.go:88); >= 1 disables the cut */
    float temp;            /* > 0 (main.go:379-381 turns 0 into 0.5) */
    float repeat_penalty;  /* 1.10 (main.go:90) */
    uint64_t seed;
} lh_sample_params;
/* One call on host logits (op-level parity): last_n_tokens = the ring contents (membership only, order free).
 * cand_ids / cand_probs / n_keep (optional, top_k entries) return the kept candidates in rank order after the
 * topP rescale, i.e. the reference's logitsID / probs right before the random pick. */
int lh_sample_top_p_top_k(lh_ctx* ctx, const float* logits, uint32_t n_logits, const uint32_t* last_n_tokens, uint32_t n_last,
                          const lh_sample_params* sp, uint64_t draw, uint32_t* token_out,
                          uint32_t* cand_ids, float* cand_probs, uint32_t* n_keep);
/* The generation loop of server.Do (server.go:127-217) resident on the device: ring of ring_size zeros
 * (server.go:127-138; the reference uses CtxSize), prompt ids appended and evaluated in one Eval, then
 * n_predict x { sample -> append -> Eval(N = 1) } with no Eval after the last sample.  Sampling call s uses draw = s.
 * No host round trip per token; out_tokens gets the n_predict sampled ids. */
int lh_llama_decode_sample(lh_llama* m, const uint32_t* prompt, uint32_t n_prompt, uint32_t n_predict, uint32_t ring_size,
                           const lh_sample_params* sp, uint32_t* out_tokens);

/* ---- pods in ONE weight pass (batched decode) ---------------------------------------------------------------------
 * The reference's only parallelism is request-level: Engine() starts up to MaxPods concurrent Do() goroutines
 * (pkg/server/server.go:84-106), each with its own llama.Context over the shared Model (server.go:151).  On the CPU they share
 * the cores.  On the GPU every N = 1 Eval streams all weights, so P pods decoding on their own read P x 26.4 GB per round of
 * tokens for the bandwidth of one.  An lh_batch binds the stages of P streams (lh_llama_create with the SAME weights and layer
 * range and ONE KV cache each, all on one lh_ctx) and evaluates a decode step of all of them as one P-row pass: weights read
 * once, RoPE / cache append / attention per row at the row's own position of its own cache.  A row's result does not depend on
 * its index or on the other rows.  Shapes the P-row kernels are not built for run row by row (same results, P passes):
 * lh_batch_batched() tells.  1 <= P <= 64.  The pods must outlive the batch; while it exists use them only through it or for
 * whole-prompt Evals (lh_llama_stage / lh_llama_eval with n > 1 rows).
 * Token ids, positions and residual rows of a tick live at fixed device addresses and the tick is ONE captured hipGraph that
 * also moves the positions on: a tick costs the host one graph launch whatever the stage's length. */
typedef struct lh_batch lh_batch;
int lh_batch_create(lh_ctx* ctx, lh_llama* const* pods, uint32_t n_pods, lh_batch** out);
void lh_batch_destroy(lh_batch* b);
uint32_t lh_batch_rows(const lh_batch* b);
int lh_batch_batched(const lh_batch* b);          /* 1: the rows share one weight pass; 0: row by row */
/* Device arrays a transport delivers into / reads from: [rows] ids evaluated by the next tick (first stage) and [rows] ids the
 * last tick produced (last stage).  On a whole-model batch a tick feeds the second into the first itself. */
uint32_t* lh_batch_tokens_dev(lh_batch* b);
uint32_t* lh_batch_ids_dev(lh_batch* b);
int lh_batch_read_ids(lh_batch* b, uint32_t* ids_host);   /* waits for the stream, then copies the [rows] produced ids to the host */
/* Position (= llama.Eval's pastCount) of every row's next token and, tokens != NULL, the ids themselves (first stage), from the
 * host; the rows' output lists restart. */
int lh_batch_set(lh_batch* b, const uint32_t* tokens_or_null, const uint32_t* past);
/* server.Do's prompt Eval (server.go:185-192) for every row, each on its own cache at position 0: prompts[i][0..n_prompt[i])
 * (host; first stage only, n_prompt on every stage).  Later stages read / earlier stages write the rows of all prompts one after
 * the other in x_in_dev / x_out_dev (sum(n_prompt) x embd floats).  The last stage leaves each row's next id (argmax of its last
 * prompt row, or the first sampler draw) in lh_batch_ids_dev; every stage then stands at position n_prompt[i] per row. */
int lh_batch_prompt(lh_batch* b, const uint32_t* const* prompts, const uint32_t* n_prompt, const float* x_in_dev, float* x_out_dev);
/* One decode tick for every row (llama.Eval with N = 1 per stream, llama.go:211-426): x_in_dev / x_out_dev = [rows][embd]
 * residual rows from the previous / for the next stage (NULL on the first / last stage).  Optional device copies of the
 * [rows][vocab] logits and the [rows] produced ids.  Asynchronous on the context's stream; positions advance by one. */
int lh_batch_stage(lh_batch* b, const float* x_in_dev, float* x_out_dev, float* logits_dev, uint32_t* ids_dev);
/* From now on the last stage draws every row's id with SampleTopPTopK (llama.go:455-707; same device sampler and counter-based
 * uniforms as lh_llama_decode_sample, every row seeded like a solo run) instead of the argmax.  ring_init[i][0..n_init[i]) = ids
 * already appended to row i's lastNTokens ring of ring_size slots (the prompt, server.go:193-197).  sp = NULL: back to greedy. */
int lh_batch_set_sampler(lh_batch* b, const lh_sample_params* sp, uint32_t ring_size, const uint32_t* const* ring_init, const uint32_t* n_init);
/* Whole-model pods: n_steps device-resident ticks from (first_tokens[i], past[i]); out_tokens[i * n_steps + s] = id row i
 * produced at step s; logits_last_host (optional) = [rows][vocab] logits of the final tick.
 * Context swap: a tick (here and in lh_batch_stage) of a WHOLE-MODEL batch whose row stands at the window's end first swaps that row's
 * context as server.Do does (server.go:160-172, lh_llama_set_keep per pod): the re-fed run is evaluated as one Eval on the row's own
 * cache, then the tick takes the row's pending token behind it.  It needs the tokens of the row's window, which the batch knows when the
 * row's prompt ran through lh_batch_prompt (or its Evals through its lh_llama with host ids); otherwise, and for a stage of a layer shard,
 * such a tick fails with LH_EINVAL before anything is enqueued. */
int lh_batch_decode(lh_batch* b, const uint32_t* first_tokens, const uint32_t* past, uint32_t n_steps, uint32_t* out_tokens, float* logits_last_host);

/* ---- multi-GPU: layer shard over RCCL point-to-point ------------------------------------------------
 * The reference's only parallel dimension is request-level "pods" (pkg/server/server.go:84-106: Engine() starts up to
 * MaxPods concurrent Do() goroutines; server.go:151: each with its own llama.Context over the shared Model).  Layers
 * shard in contiguous blocks over one process per GPU; the fp32 residual stream [n x embd] hops rank r -> r+1 and the
 * sampled token id returns from the last rank to rank 0, both as RCCL send/recv over xGMI, issued HERE (on the context's
 * stream, grouped) so that a Go host needs no collective library of its own: it only moves the 128-byte unique id from
 * rank 0 to the other ranks over whatever channel it already has (a file, a socket, its job queue).
 * librccl.so.1 is loaded on first use (dlopen), so single-GPU hosts never need it. */
#define LH_COMM_ID_BYTES 128
typedef struct lh_comm lh_comm;
int lh_comm_unique_id(lh_ctx* ctx, uint8_t id[LH_COMM_ID_BYTES]);       /* ncclGetUniqueId: call on rank 0, distribute */
int lh_comm_init(lh_ctx* ctx, int rank, int world, const uint8_t id[LH_COMM_ID_BYTES], lh_comm** out); /* ncclCommInitRank on ctx's device */
/* Alternative transport for hosts without RCCL peers (tests with several ranks on ONE GPU, which RCCL refuses; TCP):
 * the library stages the messages through host memory and hands both directions of a tick to ONE call, which must
 * post the send and the receive together (a ring of blocking sends would wait on itself).  NULL buffers = no message. */
/* ABI note: lh_comm_init_hooks copies the WHOLE struct and lh_comm_abort calls `abort` whenever it is non-NULL: zero-initialise the struct
 * (`lh_comm_hooks h = {0};`) and set the members you provide - a host that fills only `user` / `exchange` of an uninitialised struct
 * hands the library an indeterminate function pointer.  The struct's size is part of ABI version 1 (lh_abi_version). */
typedef struct lh_comm_hooks {
    void* user;
    int (*exchange)(void* user, const void* send_host, uint64_t send_bytes, int send_peer, void* recv_host, uint64_t recv_bytes, int recv_peer);
    /* optional (may be NULL): lh_comm_abort on this transport - make what the PEERS have pending against this rank fail instead of
     * block (the counterpart of ncclCommAbort: e.g. send them a message of the wrong size, or close the connection) */
    void (*abort)(void* user);
} lh_comm_hooks;
int lh_comm_init_hooks(lh_ctx* ctx, int rank, int world, const lh_comm_hooks* hooks, lh_comm** out);
void lh_comm_destroy(lh_comm* comm);
/* Tear the communicator down without waiting for the peers (ncclCommAbort): what they have pending against this rank fails instead
 * of blocking.  Used by lh_pipeline_run when a rank fails mid-run; afterwards only lh_comm_destroy is valid. */
int lh_comm_abort(lh_comm* comm);
int lh_comm_rank(const lh_comm* comm);
int lh_comm_world(const lh_comm* comm);
/* One grouped send + receive of device buffers, asynchronous on the context's stream (ncclGroupStart / ncclSend /
 * ncclRecv / ncclGroupEnd).  Either side may be absent (NULL / 0 bytes). */
int lh_comm_exchange(lh_comm* comm, const void* send_dev, uint64_t send_bytes, int send_peer, void* recv_dev, uint64_t recv_bytes, int recv_peer);

/* Pods as pipeline streams: the schedule that keeps every rank busy lives here, not in the host.
 * A "unit" is one Eval per stream (the prompt, or one decode step).  With Q = max(pods, world), rank r evaluates stream
 * p at unit u in tick t = u*Q + p + r; after every tick all ranks exchange once (ring shift): rank r sends what it just
 * produced to r+1 (the last rank sends the token id to rank 0) and receives what r-1 produced in the same tick.
 * pods >= world fills the pipeline (aggregate throughput); pods = 1 is the single greedy stream walking through the
 * stages (latency curve).  lh_pipeline_schedule is the pure function (no GPU): ticks as seen by `rank`, -1 = idle. */
typedef struct lh_tick { uint32_t t; int32_t stream, unit, recv_stream, recv_unit; } lh_tick;
int lh_pipeline_schedule(uint32_t rank, uint32_t world, uint32_t pods, uint32_t units, lh_tick* out, uint32_t cap); /* returns the tick count */
/* The scheduler loop itself with caller-supplied stage / exchange actions (what lh_pipeline_run executes with
 * lh_llama_stage and lh_comm_exchange plugged in): lets a host, or a CPU test over another transport, drive it. */
typedef struct lh_pipeline_hooks {
    void* user;
    int (*stage)(void* user, uint32_t stream, uint32_t unit);
    int (*exchange)(void* user, int32_t send_stream, int32_t send_unit, int32_t recv_stream, int32_t recv_unit);
} lh_pipeline_hooks;
int lh_pipeline_run_hooks(uint32_t rank, uint32_t world, uint32_t pods, uint32_t units, const lh_pipeline_hooks* hooks);

typedef struct lh_pipeline lh_pipeline;
/* pods[i]: this rank's stage of stream i (lh_llama_create with the rank's [layer0, layer1) and the stream's own KV
 * cache), all on `ctx` (one stream orders compute and p2p).  comm may be NULL when the model is not sharded.
 * The streams are dealt into G = min(pods, world) groups (more when a group would exceed the rows one weight pass takes: 64 fp32,
 * 64 block-int8 too since round 4); a group is an lh_batch - its streams advance together in ONE pass over the rank's weights - and the schedule
 * above runs over groups instead of single streams.  pods = 4 world: every tick evaluates 4 rows.  max_rows_per_tick (grouped
 * variant; 0 = as many as fit) bounds the rows of a group: 1 puts every stream into its own tick (one weight pass per stream). */
int lh_pipeline_create(lh_ctx* ctx, lh_comm* comm, lh_llama* const* pods, uint32_t n_pods, lh_pipeline** out);
int lh_pipeline_create_grouped(lh_ctx* ctx, lh_comm* comm, lh_llama* const* pods, uint32_t n_pods, uint32_t max_rows_per_tick, lh_pipeline** out);
void lh_pipeline_destroy(lh_pipeline* pl);
uint32_t lh_pipeline_groups(const lh_pipeline* pl);
/* The pure function behind the grouping (no GPU): groups for `pods` streams on `world` ranks with at most max_rows_per_tick (0 = 64)
 * streams per group; stream p belongs to group p * groups / pods. */
uint32_t lh_pipeline_group_count(uint32_t pods, uint32_t world, uint32_t max_rows_per_tick);
/* server.Do for every stream at once, greedy: if n_prompt != NULL, unit 0 evaluates prompts[i][0..n_prompt[i]) at
 * position 0 (prompts is read on rank 0 only; n_prompt on every rank); then `steps` decode units follow, each feeding
 * the argmax of the previous unit.  State (position, next token) persists across calls, so run(prompts, n, W) followed
 * by run(NULL, NULL, K) continues the same streams.  Returns after the rank's stream has drained. */
int lh_pipeline_run(lh_pipeline* pl, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps);
/* The same loop as the reference runs it (pkg/server/server.go:201-204: SampleTopPTopK after every Eval): the last rank draws every
 * stream's id with the device sampler (lh_sample_params, ring of ring_size slots seeded with the prompt ids, every stream like a solo
 * lh_llama_decode_sample with the same seed) instead of the argmax.  prompts must be given on rank 0 AND on the last rank (the
 * repeat penalty needs the ring there).  run_sample(NULL, NULL, K, sp, ring) continues sampled streams. */
int lh_pipeline_run_sample(lh_pipeline* pl, const uint32_t* const* prompts, const uint32_t* n_prompt, uint32_t steps, const lh_sample_params* sp, uint32_t ring_size);
/* ModelParams.KeepCount (llama.go:47) of every stream; call it with the same value on every rank.  Past the window the streams swap context
 * like server.Do (server.go:160-172): an unsharded pipeline inside its batches' ticks, a sharded one in the unit where a stream stands at the
 * window's end - its re-fed run ((ctx - keep) / 2 tokens, the pending one last) is evaluated as one Eval at position keep by every rank in turn,
 * its residual rows travelling in front of that tick's rows in the same exchange.  Rank 0 knows the tokens (its prompts + the ids it received). */
int lh_pipeline_set_keep(lh_pipeline* pl, uint32_t keep);
/* Where the time of a run goes on THIS rank (the N > 1 curve's diagnosis): with profiling on, every tick of the following runs is bracketed
 * by HIP events on the compute stream - in front of the stage, behind it, behind the exchange - and the totals accumulate: stage_ms = the
 * rank's own kernels, exchange_ms = from the end of its stage to the end of its send / receive (RCCL: includes waiting for the predecessor's
 * data, i.e. pipeline bubbles show up here).  Three event records per tick (~ 5 us): not for the timed headline run. */
typedef struct lh_pipeline_stats { uint32_t ticks; float stage_ms, exchange_ms; } lh_pipeline_stats;
int lh_pipeline_profile(lh_pipeline* pl, int on);                      /* on: also clears the totals */
int lh_pipeline_stats_read(lh_pipeline* pl, lh_pipeline_stats* out);
/* Pre-flight of the ring: microseconds per grouped send + receive of `bytes` to the successor / from the predecessor (every rank calls it
 * at the same time; 16 KB = one 7B residual row).  SURVEY 8e expects 10-20 us per hop over xGMI. */
int lh_pipeline_hop_probe(lh_pipeline* pl, uint32_t bytes, uint32_t iters, float* us_per_hop);
/* Token ids this rank knows for a stream since creation (rank 0: received from the last rank; last rank: produced). */
int lh_pipeline_tokens(lh_pipeline* pl, uint32_t pod, uint32_t* out, uint32_t cap);

/* Time each kernel class of one decode step with HIP events on the context's stream (eager launches,
 * same kernels as the replayed graph).  Writes up to cap entries; returns the number of classes. */
typedef struct lh_kernel_time { char name[48]; uint32_t launches; float total_ms; uint64_t bytes_per_launch; } lh_kernel_time;
int lh_llama_profile_decode(lh_llama* m, uint32_t token, uint32_t past, uint32_t repeats, lh_kernel_time* out, uint32_t cap);

#ifdef __cplusplus
}
#endif
#endif
