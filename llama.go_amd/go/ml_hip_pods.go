//go:build hip

// ml_hip_pods.go — the parts of the MI355X binding BEYOND ml.GraphCompute (which is ml_hip.go, the drop-in contract of SURVEY §8b):
//   - StageHIP / BatchHIP: a pod's stage (executor + KV cache in HBM) built from llama.Model's tensors, and the pods of one GPU
//     bound into ONE weight pass per decode step (lh_batch_*): what server.Engine's MaxPods concurrent Do() goroutines
//     (server.go:84-106) become on one GPU.
//   - PipelineHIP: the same streams over a layer-sharded model on N GPUs (lh_comm_* / lh_pipeline_*).
// A deployment that only wants llama.Eval on the GPU needs ml_hip.go alone.  Like ml_hip.go this file cannot be compiled in this
// repository's image (no Go toolchain); tests/test_abi.py checks every C call in both files against include/llamahip.h (names, argument
// counts, argument and struct-field types), and the C++ twin (llama.go_amd/host/llamago.cpp) exercises the same C-ABI calls in the suite.
package ml

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../lib -lllamahip
#include <stdlib.h>
#include "llamahip.h"
*/
import "C"

import (
	"fmt"
	"os"
	"unsafe"
)

// ---- pods as pipeline streams over a layer-sharded model (include/llamahip.h: lh_comm_*, lh_pipeline_*) ----------------
// One process per GPU.  Rank 0 runs the HTTP server of pkg/server; the other ranks run the same binary with --rank r and
// only ever call PipelineHIP.Run.  The 128-byte RCCL id travels over any channel the deployment already has (here: the
// caller passes it in; cmd-line, file or a TCP hello all work).
type PipelineHIP struct {
	ctx  *Context
	comm *C.lh_comm
	pl   *C.lh_pipeline
	pods []*StageHIP
}

// CommUniqueIdHIP: call on rank 0, hand the bytes to every other rank.
func CommUniqueIdHIP(ctx *Context) [C.LH_COMM_ID_BYTES]byte {
	var id [C.LH_COMM_ID_BYTES]byte
	if rc := C.lh_comm_unique_id(ctx.hip.ctx, (*C.uint8_t)(unsafe.Pointer(&id[0]))); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	return id
}

// ---- stages: what llama.Model + one pod's KV cache look like to the fused executor --------------------------------------
// Package ml cannot import package llama (llama imports ml), so the caller hands over the tensors of llama.Model
// (pkg/llama/llama.go:181-193) and llama.Layer (:128-146) in these two structs; pkg/llama gets a three-line helper that fills
// them (INTEGRATION.md §2b).  Every tensor must have been RegisterPersistent'ed (LoadModel end, llama.go:975).
type LayerWeightsHIP struct {
	AttentionNorm, WQ, WK, WV, WO, FFNNorm, W1, W2, W3 *Tensor
}
type ModelWeightsHIP struct {
	Vocab, Embd, Heads, Layers, FF uint32 // llama.HParams (llama.go:149-158) + ffSize (llama.go:761)
	TokEmbeddings, Norm, Output    *Tensor // may be nil on ranks that do not hold them (first / last stage only)
	Layer                          []LayerWeightsHIP // len == Layers; entries outside this rank's [layer0, layer1) may be zero
}

// StageHIP is one pod's stage on this rank: the lh_llama handle plus the pod's KV cache buffers (llama.Context's kvSelf,
// llama.go:91-98, for the rank's layers only), which live in HBM and nowhere else.
type StageHIP struct {
	ctx  *Context
	h    *C.lh_llama
	k, v C.lh_buf
}

func bufOf(t *Tensor) C.lh_buf {
	if t == nil {
		return 0
	}
	b, ok := lookupPersistent(t)
	if !ok {
		fmt.Printf("\n[HALT] HIP backend: a model tensor was not registered (RegisterPersistent after LoadModel)")
		os.Exit(1)
	}
	return b
}

// NewStageHIP mirrors llama.NewContext (llama.go:91-103) for layers [layer0, layer1) of the model: a zero-filled KV cache of
// embd * (layer1 - layer0) * ctxSize floats per tensor, created directly in HBM (host pointer nil), and the executor's
// description of the stage (lh_llama_desc).  layer1 == 0 means "to the last layer".  C++ twin: make_stage, host/llamago.cpp.
func NewStageHIP(ctx *Context, w *ModelWeightsHIP, layer0, layer1, ctxSize uint32) *StageHIP {
	if layer1 == 0 {
		layer1 = w.Layers
	}
	st := &StageHIP{ctx: ctx}
	// ml.Tensor.NE is uint32 (ml.go:187): the element count of a KV cache must fit it (C++ twin: llamago_NewBatch / make_stage halt likewise)
	kvn := uint64(w.Embd) * uint64(layer1-layer0) * uint64(ctxSize)
	if layer1 <= layer0 || kvn == 0 || kvn > 0xFFFFFFFF {
		fmt.Printf("\n[HALT] NewStageHIP: KV cache of %d elements (embd %d x %d layers x ctx %d) outside uint32", kvn, w.Embd, layer1-layer0, ctxSize)
		os.Exit(1)
	}
	ne := [4]C.uint32_t{C.uint32_t(kvn), 1, 1, 1}
	if rc := C.lh_tensor_register(ctx.hip.ctx, 0, C.int(TYPE_F32), &ne[0], 1, nil, &st.k); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	if rc := C.lh_tensor_register(ctx.hip.ctx, 0, C.int(TYPE_F32), &ne[0], 1, nil, &st.v); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	// the layer table goes through C memory: a Go struct handed to C must not contain Go pointers
	n := int(w.Layers)
	layers := (*[1 << 16]C.lh_llama_layer)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.lh_llama_layer{}))))[:n:n]
	defer C.free(unsafe.Pointer(&layers[0]))
	for i := int(layer0); i < int(layer1); i++ {
		l := &w.Layer[i]
		layers[i] = C.lh_llama_layer{attention_norm: bufOf(l.AttentionNorm), wq: bufOf(l.WQ), wk: bufOf(l.WK), wv: bufOf(l.WV), wo: bufOf(l.WO),
			ffn_norm: bufOf(l.FFNNorm), w1: bufOf(l.W1), w2: bufOf(l.W2), w3: bufOf(l.W3)}
	}
	var d C.lh_llama_desc
	d.vocab, d.embd, d.heads, d.layers, d.ff, d.ctx = C.uint32_t(w.Vocab), C.uint32_t(w.Embd), C.uint32_t(w.Heads), C.uint32_t(w.Layers), C.uint32_t(w.FF), C.uint32_t(ctxSize)
	d.layer0, d.layer1 = C.uint32_t(layer0), C.uint32_t(layer1)
	if layer0 == 0 {
		d.tok_embeddings = bufOf(w.TokEmbeddings)
	}
	if layer1 == w.Layers {
		d.norm, d.output = bufOf(w.Norm), bufOf(w.Output)
	}
	d.layer = &layers[0]
	d.k_cache, d.v_cache = st.k, st.v
	d.weight_dtype = C.int(TYPE_F32)
	if rc := C.lh_llama_create(ctx.hip.ctx, &d, &st.h); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	return st
}

// Release mirrors (*llama.Context).ReleaseContext (llama.go:105-113) for the stage: the executor state and the pod's KV cache
// leave HBM.
func (st *StageHIP) Release() {
	C.lh_llama_destroy(st.h)
	C.lh_buf_free(st.ctx.hip.ctx, st.k)
	C.lh_buf_free(st.ctx.hip.ctx, st.v)
	st.h = nil
}

// ---- the pods of ONE GPU in one weight pass (include/llamahip.h: lh_batch_*) --------------------------------------------
// server.Engine starts up to MaxPods concurrent Do() goroutines over one Model (server.go:84-106, 151).  With UseHIP the engine
// instead keeps ONE BatchHIP per GPU: every pod is a row; Prompt() evaluates the pods' prompts, each Tick() advances every pod by
// one token in ONE pass over the weights (4-16 pods cost about what one costs: the decode step is bound by the weight stream).
type BatchHIP struct {
	ctx    *Context
	b      *C.lh_batch
	stages []*StageHIP
}

func NewBatchHIP(ctx *Context, stages []*StageHIP) *BatchHIP {
	n := len(stages)
	if n == 0 || n > 64 { // lh_batch_create would refuse it; the slice expression below must not panic first
		fmt.Printf("\n[HALT] NewBatchHIP: %d pods outside 1..64", n)
		os.Exit(1)
	}
	hs := (*[1 << 16]*C.lh_llama)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	defer C.free(unsafe.Pointer(&hs[0]))
	for i, st := range stages {
		hs[i] = st.h
	}
	bt := &BatchHIP{ctx: ctx, stages: stages}
	if rc := C.lh_batch_create(ctx.hip.ctx, (**C.lh_llama)(unsafe.Pointer(&hs[0])), C.uint32_t(n), &bt.b); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	return bt
}

// cPrompts copies [][]uint32 into C memory (pointer table + rows); the returned func frees it.
func cPrompts(prompts [][]uint32) (**C.uint32_t, *C.uint32_t, func()) {
	n := len(prompts)
	if n == 0 { // (SetSampler(nil prompts): nothing to hand over; &ptrs[0] below would panic)
		return nil, nil, func() {}
	}
	ptrs := (*[1 << 16]*C.uint32_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	lens := (*[1 << 16]C.uint32_t)(C.malloc(C.size_t(4 * n)))[:n:n]
	for i, pr := range prompts {
		ptrs[i] = (*C.uint32_t)(C.malloc(C.size_t(4 * (len(pr) + 1))))
		if len(pr) > 0 {
			C.memcpy(unsafe.Pointer(ptrs[i]), unsafe.Pointer(&pr[0]), C.size_t(4*len(pr)))
		}
		lens[i] = C.uint32_t(len(pr))
	}
	return (**C.uint32_t)(unsafe.Pointer(&ptrs[0])), &lens[0], func() {
		for i := range ptrs {
			C.free(unsafe.Pointer(ptrs[i]))
		}
		C.free(unsafe.Pointer(&ptrs[0]))
		C.free(unsafe.Pointer(&lens[0]))
	}
}

// Prompt: server.Do's prompt Eval (server.go:185-192) for every pod; returns the id each pod's prompt produced (greedy, or the
// first sampler draw after SetSampler).
func (bt *BatchHIP) Prompt(prompts [][]uint32) []uint32 {
	pp, nn, free := cPrompts(prompts)
	defer free()
	if rc := C.lh_batch_prompt(bt.b, pp, nn, nil, nil); rc != 0 {
		hipHalt(bt.ctx.hip.ctx)
	}
	return bt.ids()
}

// Tick: one decode step of every pod (llama.Eval with N = 1 per pod, llama.go:211-426) in one pass over the weights; returns
// the ids produced.  The ids feed the next Tick on the device; the host only reads them.
func (bt *BatchHIP) Tick() []uint32 {
	if rc := C.lh_batch_stage(bt.b, nil, nil, nil, nil); rc != 0 {
		hipHalt(bt.ctx.hip.ctx)
	}
	return bt.ids()
}

func (bt *BatchHIP) ids() []uint32 {
	out := make([]uint32, len(bt.stages))
	if rc := C.lh_batch_read_ids(bt.b, (*C.uint32_t)(unsafe.Pointer(&out[0]))); rc != 0 {
		hipHalt(bt.ctx.hip.ctx)
	}
	return out
}

// SetSampler: from the next Prompt on, ids are drawn with SampleTopPTopK (llama.go:455-707; server.go:201-204) on the device.
// prompts seed every pod's lastNTokens ring (server.go:193-197); ringSize = CtxSize in the reference (server.go:127).
func (bt *BatchHIP) SetSampler(topK uint32, topP, temp, repeatPenalty float32, seed uint64, ringSize uint32, prompts [][]uint32) {
	sp := C.lh_sample_params{top_k: C.uint32_t(topK), top_p: C.float(topP), temp: C.float(temp), repeat_penalty: C.float(repeatPenalty), seed: C.uint64_t(seed)}
	pp, nn, free := cPrompts(prompts)
	defer free()
	if rc := C.lh_batch_set_sampler(bt.b, &sp, C.uint32_t(ringSize), pp, nn); rc != 0 {
		hipHalt(bt.ctx.hip.ctx)
	}
}

// SetKeepCount: ModelParams.KeepCount (llama.go:47) of every pod.  A Tick of a pod that stands at the end of its window swaps its context
// as server.Do does (server.go:160-172) inside lh_batch_stage: the host loop needs no swap code of its own.
func (bt *BatchHIP) SetKeepCount(keep uint32) {
	for _, st := range bt.stages {
		C.lh_llama_set_keep(st.h, C.uint32_t(keep))
	}
}

func (bt *BatchHIP) Release() { C.lh_batch_destroy(bt.b) }

// NewPipelineHIP: `stages[i]` is this rank's stage of stream i (NewStageHIP over the rank's layer range: the stream's own KV
// cache, all on ctx).  world == 1 needs no communicator.  The streams advance in groups: one pass over the rank's weights per
// group and tick (lh_pipeline_create).
func NewPipelineHIP(ctx *Context, rank, world int, id [C.LH_COMM_ID_BYTES]byte, stages []*StageHIP) *PipelineHIP {
	n := len(stages)
	p := &PipelineHIP{ctx: ctx, pods: stages}
	if world > 1 {
		if rc := C.lh_comm_init(ctx.hip.ctx, C.int(rank), C.int(world), (*C.uint8_t)(unsafe.Pointer(&id[0])), &p.comm); rc != 0 {
			hipHalt(ctx.hip.ctx)
		}
	}
	hs := (*[1 << 16]*C.lh_llama)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n] // C memory: the handles are C pointers, the table must be too
	defer C.free(unsafe.Pointer(&hs[0]))
	for i, st := range stages {
		hs[i] = st.h
	}
	if rc := C.lh_pipeline_create(ctx.hip.ctx, p.comm, (**C.lh_llama)(unsafe.Pointer(&hs[0])), C.uint32_t(n), &p.pl); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	return p
}

// RunSample is Run with the reference's sampler after every Eval (server.go:201-204) instead of the argmax.  prompts must be
// given on rank 0 and on the last rank (the repeat penalty runs over the ring of prompt ids there).
func (p *PipelineHIP) RunSample(prompts [][]uint32, steps int, topK uint32, topP, temp, repeatPenalty float32, seed uint64, ringSize uint32) {
	sp := C.lh_sample_params{top_k: C.uint32_t(topK), top_p: C.float(topP), temp: C.float(temp), repeat_penalty: C.float(repeatPenalty), seed: C.uint64_t(seed)}
	if prompts == nil {
		if rc := C.lh_pipeline_run_sample(p.pl, nil, nil, C.uint32_t(steps), &sp, C.uint32_t(ringSize)); rc != 0 {
			hipHalt(p.ctx.hip.ctx)
		}
		return
	}
	pp, nn, free := cPrompts(prompts)
	defer free()
	if rc := C.lh_pipeline_run_sample(p.pl, pp, nn, C.uint32_t(steps), &sp, C.uint32_t(ringSize)); rc != 0 {
		hipHalt(p.ctx.hip.ctx)
	}
}

// Run: prompts != nil starts every stream from its prompt (server.go:185-192 feeds the prompt as one Eval), then `steps`
// greedy decode steps per stream; prompts == nil continues.  Schedule, stages and RCCL p2p all run below this call.
func (p *PipelineHIP) Run(prompts [][]uint32, steps int) {
	if prompts == nil {
		if rc := C.lh_pipeline_run(p.pl, nil, nil, C.uint32_t(steps)); rc != 0 {
			hipHalt(p.ctx.hip.ctx)
		}
		return
	}
	n := len(prompts)
	ptrs := (*[1 << 16]*C.uint32_t)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n] // C memory: no Go pointer to Go pointer
	lens := make([]C.uint32_t, n)
	for i, pr := range prompts {
		ptrs[i] = (*C.uint32_t)(C.malloc(C.size_t(4 * len(pr))))
		C.memcpy(unsafe.Pointer(ptrs[i]), unsafe.Pointer(&pr[0]), C.size_t(4*len(pr)))
		lens[i] = C.uint32_t(len(pr))
	}
	rc := C.lh_pipeline_run(p.pl, (**C.uint32_t)(unsafe.Pointer(&ptrs[0])), &lens[0], C.uint32_t(steps))
	for i := range ptrs {
		C.free(unsafe.Pointer(ptrs[i]))
	}
	C.free(unsafe.Pointer(&ptrs[0]))
	if rc != 0 {
		hipHalt(p.ctx.hip.ctx)
	}
}

// SetKeepCount: ModelParams.KeepCount (llama.go:47) of every stream; every rank calls it with the same value before the prompts.  A stream
// that stands at the end of its window is swapped inside Run as server.Do does (server.go:160-172), on all ranks in the same tick.
func (p *PipelineHIP) SetKeepCount(keep uint32) {
	if rc := C.lh_pipeline_set_keep(p.pl, C.uint32_t(keep)); rc != 0 {
		hipHalt(p.ctx.hip.ctx)
	}
}

// Profile / Stats: where the time of the following Runs goes on THIS rank - its own kernels per tick (ms) and from the end of its stage to the
// end of its send / receive (us; includes waiting for the predecessor).  Profile(true) also clears the totals.  Not for a timed run.
func (p *PipelineHIP) Profile(on bool) {
	v := C.int(0)
	if on {
		v = 1
	}
	if rc := C.lh_pipeline_profile(p.pl, v); rc != 0 {
		hipHalt(p.ctx.hip.ctx)
	}
}

func (p *PipelineHIP) Stats() (ticks uint32, stageMsPerTick, exchangeUsPerTick float32) {
	var st C.lh_pipeline_stats
	if rc := C.lh_pipeline_stats_read(p.pl, &st); rc != 0 {
		hipHalt(p.ctx.hip.ctx)
	}
	if st.ticks == 0 {
		return 0, 0, 0
	}
	return uint32(st.ticks), float32(st.stage_ms) / float32(st.ticks), float32(st.exchange_ms) * 1000 / float32(st.ticks)
}

// HopProbe: microseconds per grouped send + receive of `rows` residual rows round the ring (every rank calls it at the same time).
func (p *PipelineHIP) HopProbe(rows, embd uint32, iters int) float32 {
	var us C.float
	if rc := C.lh_pipeline_hop_probe(p.pl, C.uint32_t(rows*embd*4), C.uint32_t(iters), &us); rc != 0 {
		hipHalt(p.ctx.hip.ctx)
	}
	return float32(us)
}

// Tokens: ids of a stream known to this rank (rank 0: everything generated so far).
func (p *PipelineHIP) Tokens(pod int) []uint32 {
	n := int(C.lh_pipeline_tokens(p.pl, C.uint32_t(pod), nil, 0))
	if n <= 0 {
		return nil
	}
	out := make([]uint32, n)
	C.lh_pipeline_tokens(p.pl, C.uint32_t(pod), (*C.uint32_t)(unsafe.Pointer(&out[0])), C.uint32_t(n))
	return out
}

func (p *PipelineHIP) Release() {
	C.lh_pipeline_destroy(p.pl)
	if p.comm != nil {
		C.lh_comm_destroy(p.comm)
	}
}
