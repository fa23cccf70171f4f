//go:build hip

// ml_hip.go — the reference-side binding of the MI355X backend.  Drop this file into pkg/ml of
// gotzmann/llama.go and build with `CGO_ENABLED=1 go build -tags hip` (the stock Makefile sets
// CGO_ENABLED=0, Makefile:25; add a sibling target).  It cannot be compiled in this repository's
// image (no Go toolchain): it is deliberately logic-free — every decision lives behind the C-ABI
// of include/llamahip.h, which the C++ twin of this file (llama.go_amd/host/llamago.cpp) exercises
// in the test-suite (each fix made here is mirrored and tested there: concurrent pods, context
// create/destroy returning device memory, empty tensors).
//
// What it does:
//   - Context gains `UseHIP bool`, `HIPLastRowLogits bool` and `hip *hipState` (INTEGRATION.md hunk 1), routed exactly
//     like UseAVX/UseNEON (Options -> ModelParams llama.go:38-39 -> ml.Context ml.go:52-53).
//   - RegisterPersistent() copies a weight / KV-cache tensor to HBM once (LoadModel end, llama.go:975, through the
//     package-level model context; NewContext, llama.go:91-98, through the pod's own context).  The Go slice may then
//     be dropped.  UnregisterPersistent() releases a pod's KV caches when the pod ends (server.go:151 creates a
//     context per job: without it every job would leak 2 x embd*layers*ctx floats of HBM).
//   - GraphCompute (ml.go:1411) starts with `if ctx.UseHIP { hipGraphCompute(ctx, graph); return }`.
//     hipGraphCompute flattens Graph.Leafs/Graph.Nodes into []C.lh_tensor (ml.Tensor 1:1, with the
//     slice aliasing made explicit as storage index + float offset), calls lh_graph_compute ONCE,
//     and copies the graph's root results back into their Data slices, so llama.Eval's logits read
//     (llama.go:394-401) works unchanged.
//   - PipelineHIP: pods as pipeline streams over a layer-sharded model (lh_comm_* / lh_pipeline_*): what
//     server.Engine's MaxPods concurrent Do() goroutines (server.go:84-106) become on N GPUs.
package ml

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../lib -lllamahip
#include <stdlib.h>
#include <string.h>
#include "llamahip.h"
*/
import "C"

import (
	"fmt"
	"os"
	"sync"
	"unsafe"
)

type hipState struct {
	ctx *C.lh_ctx
	// persistent tensors registered through THIS context (a pod's KV caches): released with it
	owned []*float32
}

func hipHalt(ctx *C.lh_ctx) {
	fmt.Printf("\n[HALT] HIP backend: %s", C.GoString(C.lh_last_error(ctx))) // same print-and-exit as ml.go:1538-1539
	os.Exit(1)
}

// NewContextHIP is NewContext (ml.go:59-74) for the HIP backend: no worker goroutines are needed.
func NewContextHIP(device int) *Context {
	var c *C.lh_ctx
	if rc := C.lh_ctx_create(C.int(device), nil, &c); rc != 0 {
		hipHalt(nil)
	}
	return &Context{UseHIP: true, hip: &hipState{ctx: c}, Allocator: NewAllocator()}
}

// ReleaseContextHIP mirrors (*Context).ReleaseContext (ml.go:77-80): the pod's KV caches leave HBM with it.
func (ctx *Context) ReleaseContextHIP() {
	for _, key := range ctx.hip.owned {
		unregisterKey(ctx, key)
	}
	ctx.hip.owned = nil
	C.lh_ctx_destroy(ctx.hip.ctx)
	ctx.hip.ctx = nil
}

// The model is loaded before any pod exists (llama.go:975 runs at start-up, server.go:45 shares the Model): weights are
// registered through one package-level context on the chosen device.
var (
	modelOnce sync.Once
	modelCtx  *Context
)

func ModelContextHIP(device int) *Context {
	modelOnce.Do(func() { modelCtx = NewContextHIP(device) })
	return modelCtx
}

// persistent[&Data[0]] = device buffer.  Pods run concurrently (server.go:88-101: one goroutine per job), so the map is
// guarded: Go aborts the process on an unsynchronised concurrent map read + write.
var (
	persistentMu sync.RWMutex
	persistent   = map[*float32]C.lh_buf{}
)

func lookupPersistent(t *Tensor) (C.lh_buf, bool) {
	if len(t.Data) == 0 { // &t.Data[0] would panic on an empty slice
		return 0, false
	}
	persistentMu.RLock()
	buf, ok := persistent[&t.Data[0]]
	persistentMu.RUnlock()
	return buf, ok
}

// RegisterPersistent uploads t.Data to HBM under a stable key (the address of its backing array).  Tensors registered
// through a pod's context (its KV caches) are released by ReleaseContextHIP; weights registered through
// ModelContextHIP live as long as the process, like the reference's Model.
func RegisterPersistent(ctx *Context, t *Tensor) {
	if len(t.Data) == 0 {
		return
	}
	key := &t.Data[0]
	persistentMu.Lock()
	defer persistentMu.Unlock()
	if _, ok := persistent[key]; ok {
		return
	}
	var buf C.lh_buf
	ne := [4]C.uint32_t{C.uint32_t(t.NE[0]), C.uint32_t(t.NE[1]), C.uint32_t(t.NE[2]), C.uint32_t(t.NE[3])}
	rc := C.lh_tensor_register(ctx.hip.ctx, C.uint64_t(uintptr(unsafe.Pointer(key))), C.int(TYPE_F32), &ne[0], 1,
		unsafe.Pointer(key), &buf) // the pointee holds no Go pointers: legal for the duration of the call
	if rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	persistent[key] = buf
	if ctx != modelCtx {
		ctx.hip.owned = append(ctx.hip.owned, key)
	}
}

// UnregisterPersistent frees the device copy of a tensor (a finished pod's KV cache).
func UnregisterPersistent(ctx *Context, t *Tensor) {
	if len(t.Data) != 0 {
		unregisterKey(ctx, &t.Data[0])
	}
}

func unregisterKey(ctx *Context, key *float32) {
	persistentMu.Lock()
	buf, ok := persistent[key]
	delete(persistent, key)
	persistentMu.Unlock()
	if ok {
		C.lh_buf_free(ctx.hip.ctx, buf)
	}
}

// root follows the reference's view constructors back to the tensor that owns the bytes:
// ViewTensor users (Rope ml.go:862, Scale :948, DiagMaskInf :980, SoftMax :1005, Permute :809) and
// View1D/Reshape3D alias src0; Copy's result is a view of its destination src1 (ml.go:718).
func root(t *Tensor) *Tensor {
	for {
		switch t.op {
		case OP_VIEW, OP_RESHAPE, OP_PERMUTE, OP_TRANSPOSE, OP_ROPE, OP_SCALE, OP_DIAG_MASK_INF, OP_SOFT_MAX:
			t = t.src0
		case OP_CPY:
			t = t.src1
		default:
			return t
		}
	}
}

// viewOffset: floats between a view's first element and its owner's (0 for empty tensors).
func viewOffset(t, r *Tensor) C.uint64_t {
	if len(t.Data) == 0 || len(r.Data) == 0 {
		return 0
	}
	return C.uint64_t((uintptr(unsafe.Pointer(&t.Data[0])) - uintptr(unsafe.Pointer(&r.Data[0]))) / 4)
}

func hipGraphCompute(ctx *Context, graph *Graph) {
	nl, nn := int(graph.LeafsCount), int(graph.NodesCount)
	if nl+nn == 0 {
		return
	}
	index := make(map[*Tensor]int32, nl+nn)
	all := make([]*Tensor, 0, nl+nn)
	for i := 0; i < nl; i++ {
		index[graph.Leafs[i]] = int32(len(all))
		all = append(all, graph.Leafs[i])
	}
	// every storage owner must be visible to the C side: owners reached only through views become extra leafs
	extra := []*Tensor{}
	for i := 0; i < nn; i++ {
		if r := root(graph.Nodes[i]); r.op == OP_NONE {
			if _, ok := index[r]; !ok {
				index[r] = int32(len(all) + len(extra))
				extra = append(extra, r)
			}
		}
	}
	all = append(all, extra...)
	nl = len(all)
	for i := 0; i < nn; i++ {
		index[graph.Nodes[i]] = int32(len(all))
		all = append(all, graph.Nodes[i])
	}

	// C-owned array + staging for the small host leafs (token ids, rope/mask/scale parameters):
	// no Go pointer is ever stored in C memory.
	arr := (*[1 << 20]C.lh_tensor)(C.calloc(C.size_t(len(all)), C.size_t(unsafe.Sizeof(C.lh_tensor{}))))[:len(all):len(all)]
	defer C.free(unsafe.Pointer(&arr[0]))
	bufs := make([]C.lh_buf, len(all)) // one map lookup per tensor, under the read lock
	isPersistent := make([]bool, len(all))
	staged := 0
	for i, t := range all {
		bufs[i], isPersistent[i] = lookupPersistent(t)
		if i < nl && !isPersistent[i] {
			staged += len(t.Data)
		}
	}
	stage := (*[1 << 28]C.float)(C.malloc(C.size_t(4 * (staged + 1))))[: staged+1 : staged+1]
	defer C.free(unsafe.Pointer(&stage[0]))
	so := 0

	consumed := make([]bool, len(all))
	for i, t := range all {
		o := &arr[i]
		o.op = C.uint8_t(t.op)
		o.dtype = C.uint8_t(TYPE_F32) // Data is []float32 for every tensor, "I32" parameters included (ml.go:864-867)
		for k := 0; k < 4; k++ {
			o.ne[k] = C.uint32_t(t.NE[k])
			o.nb[k] = C.uint64_t(t.NB[k])
		}
		o.src0, o.src1 = -1, -1
		if t.src0 != nil {
			o.src0 = C.int32_t(index[t.src0])
			consumed[index[t.src0]] = true
		}
		if t.src1 != nil {
			o.src1 = C.int32_t(index[t.src1])
			consumed[index[t.src1]] = true
		}
		r := root(t)
		o.storage = C.int32_t(index[r])
		o.view_off = viewOffset(t, r)
		if r == t {
			if isPersistent[i] {
				o.buf = bufs[i]
			} else if t.op == OP_NONE && len(t.Data) > 0 {
				n := len(t.Data)
				C.memcpy(unsafe.Pointer(&stage[so]), unsafe.Pointer(&t.Data[0]), C.size_t(4*n))
				o.host = (*C.float)(unsafe.Pointer(&stage[so]))
				so += n
			}
		}
	}

	// HIPLastRowLogits: set by llama.Eval (which copies out only row N-1 of the lm_head, llama.go:394-401) so that a fused plan
	// skips the other N-1 logits rows; generic ml.Graph users leave it false and get every node in full.
	flags := C.uint32_t(0)
	if ctx.HIPLastRowLogits {
		flags = C.LH_GRAPH_LAST_ROW_LOGITS
	}
	if rc := C.lh_graph_compute(ctx.hip.ctx, &arr[0], C.uint32_t(nl), C.uint32_t(nn), flags); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}

	// graph roots (nodes nobody consumes) that do not live in a persistent buffer go back to their Data slices;
	// for llama.Eval that is exactly the lm_head output (llama.go:384-401): the K/V cache copies stay in HBM.
	for i := nl; i < len(all); i++ {
		t := all[i]
		if consumed[i] || len(t.Data) == 0 {
			continue
		}
		if _, ok := lookupPersistent(root(t)); ok {
			continue
		}
		n := C.uint64_t(t.Nelements())
		if rc := C.lh_node_read(ctx.hip.ctx, C.uint32_t(i), 0, (*C.float)(unsafe.Pointer(&t.Data[0])), n); rc != 0 {
			hipHalt(ctx.hip.ctx)
		}
	}
}

// ---- pods as pipeline streams over a layer-sharded model (include/llamahip.h: lh_comm_*, lh_pipeline_*) ----------------
// One process per GPU.  Rank 0 runs the HTTP server of pkg/server; the other ranks run the same binary with --rank r and
// only ever call PipelineHIP.Run.  The 128-byte RCCL id travels over any channel the deployment already has (here: the
// caller passes it in; cmd-line, file or a TCP hello all work).
type PipelineHIP struct {
	ctx  *Context
	comm *C.lh_comm
	pl   *C.lh_pipeline
	pods []*C.lh_llama
}

// CommUniqueIdHIP: call on rank 0, hand the bytes to every other rank.
func CommUniqueIdHIP(ctx *Context) [C.LH_COMM_ID_BYTES]byte {
	var id [C.LH_COMM_ID_BYTES]byte
	if rc := C.lh_comm_unique_id(ctx.hip.ctx, (*C.uint8_t)(unsafe.Pointer(&id[0]))); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	return id
}

// NewPipelineHIP: `stages[i]` is this rank's lh_llama of stream i (lh_llama_create over the rank's layer range and the
// stream's own KV cache, all on ctx).  world == 1 needs no communicator.
func NewPipelineHIP(ctx *Context, rank, world int, id [C.LH_COMM_ID_BYTES]byte, stages []*C.lh_llama) *PipelineHIP {
	p := &PipelineHIP{ctx: ctx, pods: stages}
	if world > 1 {
		if rc := C.lh_comm_init(ctx.hip.ctx, C.int(rank), C.int(world), (*C.uint8_t)(unsafe.Pointer(&id[0])), &p.comm); rc != 0 {
			hipHalt(ctx.hip.ctx)
		}
	}
	if rc := C.lh_pipeline_create(ctx.hip.ctx, p.comm, (**C.lh_llama)(unsafe.Pointer(&stages[0])), C.uint32_t(len(stages)), &p.pl); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	return p
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
