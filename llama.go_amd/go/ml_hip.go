//go:build hip

// ml_hip.go — the reference-side binding of the MI355X backend for ml.GraphCompute (SURVEY §8b; pods and pipelines: ml_hip_pods.go).
// Drop both files into pkg/ml of gotzmann/llama.go and build with `CGO_ENABLED=1 go build -tags hip` (the stock Makefile sets
// CGO_ENABLED=0, Makefile:25; add a sibling target).  It cannot be compiled in this repository's image (no Go toolchain): it is
// logic-free - every decision lives behind the C-ABI of include/llamahip.h, which the C++ twin (llama.go_amd/host/llamago.cpp)
// exercises in the test-suite - and tests/test_abi.py checks every C call below against the header (names, arity, types, fields).
//   - Context gains `UseHIP`, `HIPLastRowLogits bool` and `hip *hipState` (INTEGRATION.md hunk 1), routed like UseAVX/UseNEON.
//   - RegisterPersistent() copies a weight / KV-cache tensor to HBM once (llama.go:975 through the model context; llama.go:91-98
//     through the pod's own); UnregisterPersistent() / ReleaseContextHIP() release a pod's KV caches (server.go:151: a context per job).
//   - GraphCompute (ml.go:1411) starts with `if ctx.UseHIP { hipGraphCompute(ctx, graph); return }`: Graph.Leafs/Nodes flattened into
//     []C.lh_tensor (ml.Tensor 1:1, slice aliasing made explicit as storage index + float offset), ONE lh_graph_compute, and the graph's
//     root results copied back into their Data slices, so llama.Eval's logits read (llama.go:394-401) works unchanged.
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
	ctx   *C.lh_ctx
	owned []*Tensor // persistent tensors registered through THIS context (a pod's KV caches): released with it
}

func hipHalt(ctx *C.lh_ctx) { // same print-and-exit as ml.go:1538-1539
	fmt.Printf("\n[HALT] HIP backend: %s", C.GoString(C.lh_last_error(ctx)))
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
		UnregisterPersistent(ctx, key)
	}
	ctx.hip.owned = nil
	C.lh_ctx_destroy(ctx.hip.ctx)
	ctx.hip.ctx = nil
}

// TimeComputesHIP arms (or disarms) the library's timers around every GraphCompute of this context - HIP events on its stream and the host clock
// inside lh_graph_compute (SURVEY 8d config 2) - and zeroes the sums; ComputeStatsHIP returns them: calls, host microseconds, device microseconds.
func (ctx *Context) TimeComputesHIP(on int) { C.lh_ctx_time_computes(ctx.hip.ctx, C.int(on)) }
func (ctx *Context) ComputeStatsHIP() (calls uint64, wallUs, deviceUs float64) {
	var st C.lh_compute_stats
	C.lh_ctx_compute_stats(ctx.hip.ctx, &st)
	return uint64(st.calls), float64(st.wall_us), float64(st.device_us)
}

// The model is loaded before any pod exists (llama.go:975, server.go:45 shares it): weights are registered through one package-level context.
var (
	modelOnce sync.Once
	modelCtx  *Context
)

func ModelContextHIP(device int) *Context {
	modelOnce.Do(func() { modelCtx = NewContextHIP(device) })
	return modelCtx
}

// persistent[tensor] = device buffer, keyed by the *Tensor (the very objects the graph's MulMat / View1D nodes point to,
// llama.go:263-265, 274-275), not by &Data[0]: a pointer into the backing array would keep 27 GB of host weights reachable.
// Pods run concurrently (server.go:88-101), so the maps are guarded.
var (
	persistentMu sync.RWMutex
	persistent   = map[*Tensor]C.lh_buf{}
	podOwned     = map[*Tensor]bool{} // registered through a pod's context (its KV caches)
)

func lookupPersistent(t *Tensor) (C.lh_buf, bool) {
	persistentMu.RLock()
	buf, ok := persistent[t]
	persistentMu.RUnlock()
	return buf, ok
}

// ReleaseHostWeights: after RegisterPersistent the device holds the only copy inference needs; Data = nil lets the garbage collector
// free the host copy (27 GB for 7B fp32).  ONLY weights registered through the model context qualify (consumed whole by MulMat /
// GetRows / Mul, never through a view); a pod's KV cache is addressed through View1D offsets derived from its Data slice
// (viewOffset), so a tensor owned by a pod context is refused.
func ReleaseHostWeights(tensors ...*Tensor) {
	persistentMu.Lock()
	defer persistentMu.Unlock()
	for _, t := range tensors {
		if _, ok := persistent[t]; t != nil && ok && !podOwned[t] {
			t.Data = nil
		}
	}
}

// RegisterPersistent uploads t.Data to HBM.  Tensors registered through a pod's context (its KV caches) are released by
// ReleaseContextHIP; weights registered through ModelContextHIP live as long as the process, like the reference's Model.
func RegisterPersistent(ctx *Context, t *Tensor) {
	if len(t.Data) == 0 {
		return
	}
	persistentMu.Lock()
	defer persistentMu.Unlock()
	if _, ok := persistent[t]; ok {
		return
	}
	var buf C.lh_buf
	ne := [4]C.uint32_t{C.uint32_t(t.NE[0]), C.uint32_t(t.NE[1]), C.uint32_t(t.NE[2]), C.uint32_t(t.NE[3])}
	// library-side key: the tensor's address as a number; the data pointer is used for the duration of the call only (cgo rule)
	if rc := C.lh_tensor_register(ctx.hip.ctx, C.uint64_t(uintptr(unsafe.Pointer(t))), C.int(TYPE_F32), &ne[0], 1, unsafe.Pointer(&t.Data[0]), &buf); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}
	persistent[t] = buf
	if ctx != modelCtx {
		ctx.hip.owned = append(ctx.hip.owned, t)
		podOwned[t] = true
	}
}

// UnregisterPersistent frees the device copy of a tensor (a finished pod's KV cache).
func UnregisterPersistent(ctx *Context, key *Tensor) {
	persistentMu.Lock()
	buf, ok := persistent[key]
	delete(persistent, key)
	delete(podOwned, key)
	persistentMu.Unlock()
	if ok {
		C.lh_buf_free(ctx.hip.ctx, buf)
	}
}

// root follows the reference's view constructors back to the tensor that owns the bytes: ViewTensor users (Rope ml.go:862, Scale :948,
// DiagMaskInf :980, SoftMax :1005, Permute :809) and View1D/Reshape3D alias src0; Copy's result is a view of its destination src1 (:718).
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

// viewOffset: floats between a view's first element and its owner's (0 for empty tensors).  A view of an owner whose host copy is
// gone has no offset to derive: ReleaseHostWeights only accepts tensors nothing views, so that is a caller bug and halts.
func viewOffset(t, r *Tensor) C.uint64_t {
	if t != r && len(t.Data) != 0 && len(r.Data) == 0 {
		fmt.Printf("\n[HALT] HIP backend: view of a tensor whose host data was released")
		os.Exit(1)
	}
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

	// C-owned array + staging for the small host leafs (token ids, rope/mask/scale parameters): no Go pointer is ever stored in C memory
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

	// HIPLastRowLogits: set by llama.Eval (it copies out only row N-1 of the lm_head, llama.go:394-401): a fused plan skips the other rows
	flags := C.uint32_t(0)
	if ctx.HIPLastRowLogits {
		flags = C.LH_GRAPH_LAST_ROW_LOGITS
	}
	if rc := C.lh_graph_compute(ctx.hip.ctx, &arr[0], C.uint32_t(nl), C.uint32_t(nn), flags); rc != 0 {
		hipHalt(ctx.hip.ctx)
	}

	// graph roots (nodes nobody consumes) outside persistent buffers go back to their Data slices: for llama.Eval the lm_head output
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

