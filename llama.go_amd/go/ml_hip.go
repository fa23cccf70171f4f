//go:build hip

// ml_hip.go — the reference-side binding of the MI355X backend.  Drop this file into pkg/ml of
// gotzmann/llama.go and build with `CGO_ENABLED=1 go build -tags hip` (the stock Makefile sets
// CGO_ENABLED=0, Makefile:25; add a sibling target).  It cannot be compiled in this repository's
// image (no Go toolchain): it is deliberately logic-free — every decision lives behind the C-ABI
// of include/llamahip.h, which the C++ twin of this file (llama.go_amd/host/llamago.cpp) exercises
// in the test-suite.
//
// What it does:
//   - Context gains `UseHIP bool`, `HIPLastRowLogits bool` + `hip *hipState`, routed exactly like UseAVX/UseNEON
//     (Options -> ModelParams llama.go:38-39 -> ml.Context ml.go:52-53).
//   - RegisterPersistent() copies a weight / KV-cache tensor to HBM once (LoadModel end,
//     llama.go:975; NewContext, llama.go:91-98).  The Go slice may then be dropped.
//   - GraphCompute (ml.go:1411) starts with `if ctx.UseHIP { hipGraphCompute(ctx, graph); return }`.
//     hipGraphCompute flattens Graph.Leafs/Graph.Nodes into []C.lh_tensor (ml.Tensor 1:1, with the
//     slice aliasing made explicit as storage index + float offset), calls lh_graph_compute ONCE,
//     and copies the graph's root results back into their Data slices, so llama.Eval's logits read
//     (llama.go:394-401) works unchanged.
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
	"unsafe"
)

type hipState struct {
	ctx *C.lh_ctx
}

// NewContextHIP is NewContext (ml.go:59-74) for the HIP backend: no worker goroutines are needed.
func NewContextHIP(device int) *Context {
	var c *C.lh_ctx
	if rc := C.lh_ctx_create(C.int(device), nil, &c); rc != 0 {
		fmt.Printf("\n[HALT] HIP backend: %s", C.GoString(C.lh_last_error(nil)))
		os.Exit(1)
	}
	return &Context{UseHIP: true, hip: &hipState{ctx: c}, Allocator: NewAllocator()}
}

// ReleaseContextHIP mirrors (*Context).ReleaseContext (ml.go:77-80).
func (ctx *Context) ReleaseContextHIP() { C.lh_ctx_destroy(ctx.hip.ctx) }

// persistent[&Data[0]] = device buffer; weights and KV caches live here for the life of the process.
var persistent = map[*float32]C.lh_buf{}

// RegisterPersistent uploads t.Data to HBM under a stable key (the address of its backing array).
func RegisterPersistent(ctx *Context, t *Tensor) {
	key := &t.Data[0]
	if _, ok := persistent[key]; ok {
		return
	}
	var buf C.lh_buf
	ne := [4]C.uint32_t{C.uint32_t(t.NE[0]), C.uint32_t(t.NE[1]), C.uint32_t(t.NE[2]), C.uint32_t(t.NE[3])}
	rc := C.lh_tensor_register(ctx.hip.ctx, C.uint64_t(uintptr(unsafe.Pointer(key))), C.int(t.Type), &ne[0], 1,
		unsafe.Pointer(key), &buf) // the pointee holds no Go pointers: legal for the duration of the call
	if rc != 0 {
		fmt.Printf("\n[HALT] HIP backend: %s", C.GoString(C.lh_last_error(ctx.hip.ctx)))
		os.Exit(1)
	}
	persistent[key] = buf
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

func hipGraphCompute(ctx *Context, graph *Graph) {
	nl, nn := int(graph.LeafsCount), int(graph.NodesCount)
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
	staged := 0
	for i := 0; i < nl; i++ {
		if _, ok := persistent[&all[i].Data[0]]; !ok {
			staged += len(all[i].Data)
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
		o.view_off = C.uint64_t((uintptr(unsafe.Pointer(&t.Data[0])) - uintptr(unsafe.Pointer(&r.Data[0]))) / 4)
		if r == t {
			if buf, ok := persistent[&t.Data[0]]; ok {
				o.buf = buf
			} else if t.op == OP_NONE {
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
		fmt.Printf("\n[HALT] %s", C.GoString(C.lh_last_error(ctx.hip.ctx))) // same print-and-exit as ml.go:1538-1539
		os.Exit(1)
	}

	// graph roots (nodes nobody consumes) that do not live in a persistent buffer go back to their Data slices;
	// for llama.Eval that is exactly the lm_head output (llama.go:384-401): the K/V cache copies stay in HBM.
	for i := nl; i < len(all); i++ {
		t := all[i]
		if consumed[i] {
			continue
		}
		if _, ok := persistent[&root(t).Data[0]]; ok {
			continue
		}
		n := C.uint64_t(t.Nelements())
		if rc := C.lh_node_read(ctx.hip.ctx, C.uint32_t(i), 0, (*C.float)(unsafe.Pointer(&t.Data[0])), n); rc != 0 {
			fmt.Printf("\n[HALT] %s", C.GoString(C.lh_last_error(ctx.hip.ctx)))
			os.Exit(1)
		}
	}
}
