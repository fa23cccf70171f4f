/* include/llamago.h — host-side mirror of the reference's operator surface, as a C API.
 *
 * The reference (gotzmann/llama.go) is Go and there is no Go toolchain in this image, so the host
 * layer above the C-ABI boundary (include/llamahip.h) is written in C++ and exported with C
 * linkage under the reference's own names: every function below is the Go function of the same
 * name in pkg/ml/ml.go or pkg/llama/llama.go (file:line cited per entry), same argument meaning,
 * same shapes/strides/view semantics.  Two libraries export this exact API:
 *
 *   llama.go_amd/lib/libllamago.so   PRODUCT  — graph building on the host, ml_GraphCompute crosses
 *                                    the C-ABI once (lh_graph_compute) and runs on the MI355X.
 *   oracle/liboracle.so              TEST INFRASTRUCTURE ONLY — CPU restatement of the reference
 *                                    arithmetic (scalar pure-Go order, or the reference's own AVX
 *                                    vdot when useAVX=1), used by tests/ as the checker.
 *
 * so one harness (tests/, bench.py's cpu_baseline leg) drives both and compares.
 *
 * Error behaviour: the reference prints "[HALT] ..." and os.Exit(1)s on invariant violations
 * (e.g. ml.go:254-257, 2116-2124).  A library must not exit its host: constructors return NULL and
 * compute calls return non-zero, with the reference's message retrievable via ml_LastError().
 */
#ifndef LLAMAGO_H
#define LLAMAGO_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ml_context ml_context; /* ml.Context  ml.go:50-57  */
typedef struct ml_tensor ml_tensor;   /* ml.Tensor   ml.go:180-203 */
typedef struct ml_graph ml_graph;     /* ml.Graph    ml.go:31-45  */

/* ml.DType ml.go:85-94 (numeric values identical) */
enum { ML_TYPE_F32 = 0, ML_TYPE_F16 = 1, ML_TYPE_Q4_0 = 2, ML_TYPE_Q4_1 = 3, ML_TYPE_I8 = 4, ML_TYPE_I16 = 5, ML_TYPE_I32 = 6,
       ML_TYPE_Q8_0 = 7 /* ours: block-int8 weights; the reference has no storage for it */ };

/* ml.optype ml.go:133-174 (numeric values identical) */
enum { ML_OP_NONE = 0, ML_OP_DUP, ML_OP_ADD, ML_OP_SUB, ML_OP_MUL, ML_OP_DIV, ML_OP_SQR, ML_OP_SQRT, ML_OP_SUM, ML_OP_MEAN,
       ML_OP_REPEAT, ML_OP_ABS, ML_OP_SGN, ML_OP_NEG, ML_OP_STEP, ML_OP_RELU, ML_OP_GELU, ML_OP_SILU, ML_OP_NORM,
       ML_OP_RMS_NORM, ML_OP_MUL_MAT, ML_OP_SCALE, ML_OP_CPY, ML_OP_RESHAPE, ML_OP_VIEW, ML_OP_PERMUTE, ML_OP_TRANSPOSE,
       ML_OP_GET_ROWS, ML_OP_DIAG_MASK_INF, ML_OP_SOFT_MAX, ML_OP_ROPE, ML_OP_CONV_1D_1S, ML_OP_CONV_1D_2S,
       ML_OP_FLASH_ATTN, ML_OP_FLASH_FF, ML_OP_COUNT };

/* ---- context ---------------------------------------------------------------------------------- */
/* ml.NewContext ml.go:59-74.  useAVX/useNEON follow the reference's flags: the checker switches its dot-product
 * summation order on them (useAVX 1: utils/floats_avx.c 8-lane order; useNEON: utils/floats_neon.c 4-lane order;
 * neither: the scalar pure-Go order; useAVX 2, checker only: float64 accumulation, the truth leg of BASELINE.md §3).
 * The product ignores both: it always runs the HIP path. */
ml_context* ml_NewContext(int maxThreads, int useAVX, int useNEON);
void ml_ReleaseContext(ml_context* ctx); /* ml.go:77-80 */
const char* ml_LastError(void);          /* last "[HALT]"-class message of this thread, "" if none */

/* ---- tensors ---------------------------------------------------------------------------------- */
ml_tensor* ml_NewTensor1D(ml_context* ctx, int dt, uint32_t ne0);                                         /* ml.go:742 */
ml_tensor* ml_NewTensor2D(ml_context* ctx, int dt, uint32_t ne0, uint32_t ne1);                           /* ml.go:747 */
ml_tensor* ml_NewTensor3D(ml_context* ctx, int dt, uint32_t ne0, uint32_t ne1, uint32_t ne2);             /* ml.go:751 */
ml_tensor* ml_NewFP32(ml_context* ctx, float value);                                                      /* ml.go:915 */
float* ml_TensorData(ml_tensor* t);            /* Tensor.Data (host copy; device-resident for product weights after upload) */
void ml_TensorShape(const ml_tensor* t, uint32_t ne[4], uint32_t nb[4]); /* Tensor.NE / Tensor.NB  ml.go:187-188 */
int ml_TensorOp(const ml_tensor* t);
uint64_t ml_Nelements(const ml_tensor* t);     /* ml.go:217 */
/* Copy a computed tensor's elements (in its own strided layout order: contiguous tensors only) to dst.
 * Product: device -> host through lh_read; oracle: memcpy.  Valid until the next ml_GraphCompute. */
int ml_TensorRead(ml_context* ctx, ml_tensor* t, float* dst, uint64_t n);
/* Mark a leaf tensor's host data as changed so the product re-uploads it (weights are uploaded once). */
void ml_TensorDirty(ml_tensor* t);
void ml_FreeTensor(ml_tensor* t);              /* Go has a GC; C callers free leafs they created */

/* ---- operator constructors (graph nodes) -------------------------------------------------------- */
ml_tensor* ml_Add(ml_context* ctx, ml_tensor* a, ml_tensor* b);                 /* ml.go:321-360  */
ml_tensor* ml_Mul(ml_context* ctx, ml_tensor* a, ml_tensor* b);                 /* ml.go:241-287  */
ml_tensor* ml_MulMat(ml_context* ctx, ml_tensor* a, ml_tensor* b);              /* ml.go:295-318  */
ml_tensor* ml_Repeat(ml_context* ctx, ml_tensor* a, ml_tensor* b);              /* ml.go:487-513  */
ml_tensor* ml_GetRows(ml_context* ctx, ml_tensor* a, ml_tensor* b);             /* ml.go:528-557  */
ml_tensor* ml_RMSNorm(ml_context* ctx, ml_tensor* a);                           /* ml.go:559-597  */
ml_tensor* ml_View1D(ml_context* ctx, ml_tensor* a, uint32_t ne0, uint32_t offset_floats); /* ml.go:601-617 */
ml_tensor* ml_Copy(ml_context* ctx, ml_tensor* a, ml_tensor* b);                /* ml.go:700-735  */
ml_tensor* ml_Permute(ml_context* ctx, ml_tensor* a, uint32_t ax0, uint32_t ax1, uint32_t ax2, uint32_t ax3); /* ml.go:786-845 */
ml_tensor* ml_Rope(ml_context* ctx, ml_tensor* a, uint32_t past, uint32_t dims, uint32_t mode);          /* ml.go:848-880 */
ml_tensor* ml_Reshape3D(ml_context* ctx, ml_tensor* a, uint32_t ne0, uint32_t ne1, uint32_t ne2);        /* ml.go:882-912 */
ml_tensor* ml_Scale(ml_context* ctx, ml_tensor* a, ml_tensor* b);               /* ml.go:933-961  */
ml_tensor* ml_DiagMaskInf(ml_context* ctx, ml_tensor* a, uint32_t past);        /* ml.go:968-990  */
ml_tensor* ml_SoftMax(ml_context* ctx, ml_tensor* a);                           /* ml.go:993-1014 */
ml_tensor* ml_Silu(ml_context* ctx, ml_tensor* a);                              /* ml.go:1018-1043 */

/* ---- graph ---------------------------------------------------------------------------------------- */
ml_graph* ml_NewGraph(void);                                   /* &ml.Graph{}  llama.go:232 */
void ml_FreeGraph(ml_graph* g);                                /* frees the graph and every non-leaf tensor it reached */
int ml_BuildForwardExpand(ml_graph* g, ml_tensor* t);          /* ml.go:642-697 */
uint32_t ml_GraphNodesCount(const ml_graph* g);
ml_tensor* ml_GraphNode(const ml_graph* g, uint32_t i);
int ml_GraphCompute(ml_context* ctx, ml_graph* g);             /* ml.go:1411-1528 — THE call the backend replaces */

/* ---- pkg/llama ------------------------------------------------------------------------------------ */
typedef struct llama_model llama_model;     /* llama.Model   llama.go:181-193 */
typedef struct llama_context llama_context; /* llama.Context llama.go:83-88   */

typedef struct llama_hparams { /* llama.HParams llama.go:149-158 */
    uint32_t ctxSize, vocabSize, embdSize, multSize, headsCount, layersCount, rotCount, f16;
} llama_hparams;

/* llama.LoadModel llama.go:712-976: ggjt v1 .bin (f32 or f16 tensors, f16 widened to f32 at load). */
llama_model* llama_LoadModel(const char* fileName, uint32_t ctxSize);
/* Harness-only constructor (no counterpart in the reference): random-init weights of the given
 * shape from the counter-based generator specified in DESIGN.md, bit-identical in oracle and
 * product (the product generates directly in HBM).  layer range [layer0, layer1) allows a rank of a
 * layer-sharded job to materialise only its own layers (layer1 = 0 means "all"). */
llama_model* llama_NewSyntheticModel(const llama_hparams* hp, uint64_t seed, uint32_t layer0, uint32_t layer1);
/* Write the model as a ggjt v1 file (inverse of llama_LoadModel; ftype 0 = f32, 1 = f16 matrices). */
int llama_SaveModel(const llama_model* m, const char* fileName, int ftype);
void llama_FreeModel(llama_model* m);
void llama_ModelHParams(const llama_model* m, llama_hparams* out);
uint32_t llama_ModelFFSize(const llama_model* m); /* llama.go:761 */
ml_tensor* llama_ModelTensor(llama_model* m, const char* name); /* model.tensors[name] llama.go:826-861 */

llama_context* llama_NewContext(llama_model* m, uint32_t ctxSize, int maxThreads, int useAVX, int useNEON); /* llama.go:91-103 */
void llama_ReleaseContext(llama_context* lctx);                                                             /* llama.go:105-113 */
/* llama.Eval llama.go:211-426: builds the graph with the ml_* constructors above, one ml_GraphCompute,
 * copies the last row of logits to lctx.Logits.  Returns 0 (the reference always returns nil). */
int llama_Eval(llama_context* lctx, llama_model* m, const uint32_t* tokens, uint32_t n, uint32_t pastCount);
const float* llama_Logits(const llama_context* lctx); /* lctx.Logits, vocabSize floats */
const float* llama_Embedding(llama_context* c);   /* lctx.Embedding (llama.go:88, 414-419): [embd] floats, NULL unless llamago_EnableEmbedding */
ml_context* llama_MLContext(llama_context* lctx);
/* Greedy decode as defined in SURVEY §8c: argmax (lowest index on ties) of Logits after each Eval,
 * loop as server.Do (server.go:153-217): prompt in one Eval, then N=1 steps.  out_tokens gets
 * n_predict ids; if step_logits != NULL it receives n_predict*vocab floats (logits that produced each id). */
int llama_GreedyDecode(llama_context* lctx, llama_model* m, const uint32_t* prompt, uint32_t n_prompt,
                       uint32_t n_predict, uint32_t* out_tokens, float* step_logits);

/* SampleTopPTopK (llama.go:455-707).  Differences in the SIGNATURE only: the container/ring of the reference
 * (server.go:127) arrives as the array of its contents (the reference scans the whole ring, llama.go:509, so only
 * membership matters); the wall-clock seed of llama.go:658 becomes the explicit (seed, draw) pair, draw = index of
 * the sampling call; the id is returned through *token so that errors can be reported (0 = ok).  `ctx` selects the
 * device in the product and is ignored by the CPU checker.  Equal values sort by ascending id. */
int llama_SampleTopPTopK(ml_context* ctx, const float* logits, uint32_t logitsCount, const uint32_t* lastNTokens,
                         uint32_t lastNTokensSize, uint32_t topK, float topP, float temp, float repeatPenalty,
                         uint64_t seed, uint64_t draw, uint32_t* token);
/* The generation loop of server.Do (server.go:127-217) with that sampler: ring of CtxSize zeros, prompt ids appended
 * and evaluated in one Eval, then n_predict x { sample (draw = s) -> append -> Eval(N = 1) }, no Eval after the last
 * sample, no context swap (prompt + predictions must fit CtxSize). */
int llama_SampleDecode(llama_context* lctx, llama_model* m, const uint32_t* prompt, uint32_t n_prompt, uint32_t n_predict,
                       uint32_t topK, float topP, float temp, float repeatPenalty, uint64_t seed, uint32_t* out_tokens);

#ifdef __cplusplus
}
#endif
#endif
