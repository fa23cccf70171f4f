"""ctypes binding of include/llamago.h — the host-side mirror of the reference's pkg/ml + pkg/llama API.

`MLLib(path)` wraps any shared library exporting that API.  The product instance comes from
`load_product()` (libllamago.so -> libllamahip.so -> MI355X) and raises if the HIP library is absent;
the test-suite wraps its CPU checker library with the same class.  Method names are the Go
names (ml.MulMat -> MLLib.MulMat, llama.Eval -> Model/Context.Eval).
"""
import ctypes as C
import os

import numpy as np

c_u32 = C.c_uint32
c_u64 = C.c_uint64
c_f32p = C.POINTER(C.c_float)
c_u32p = C.POINTER(c_u32)
VP = C.c_void_p


class HParams(C.Structure):
    """llama.HParams (llama.go:149-158)."""
    _fields_ = [(n, c_u32) for n in ("ctxSize", "vocabSize", "embdSize", "multSize", "headsCount", "layersCount", "rotCount", "f16")]


# dtype / op numeric values (ml.go:85-94, 133-174)
TYPE_F32, TYPE_F16, TYPE_I32 = 0, 1, 6
OP_NAMES = ["NONE", "DUP", "ADD", "SUB", "MUL", "DIV", "SQR", "SQRT", "SUM", "MEAN", "REPEAT", "ABS", "SGN", "NEG", "STEP",
            "RELU", "GELU", "SILU", "NORM", "RMS_NORM", "MUL_MAT", "SCALE", "CPY", "RESHAPE", "VIEW", "PERMUTE", "TRANSPOSE",
            "GET_ROWS", "DIAG_MASK_INF", "SOFT_MAX", "ROPE", "CONV_1D_1S", "CONV_1D_2S", "FLASH_ATTN", "FLASH_FF"]


class MLError(RuntimeError):
    pass


class MLLib:
    def __init__(self, path, mode=C.RTLD_LOCAL):
        if not os.path.exists(path):
            raise MLError(f"shared library not found: {path} (run `python -c 'import __graft_entry__ as g; g.build()'`)")
        self.path = path
        self.lib = L = C.CDLL(path, mode=mode)

        def sig(name, res, *args):
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = list(args)
            return fn

        sig("ml_NewContext", VP, C.c_int, C.c_int, C.c_int)
        sig("ml_ReleaseContext", None, VP)
        sig("ml_LastError", C.c_char_p)
        sig("ml_NewTensor1D", VP, VP, C.c_int, c_u32)
        sig("ml_NewTensor2D", VP, VP, C.c_int, c_u32, c_u32)
        sig("ml_NewTensor3D", VP, VP, C.c_int, c_u32, c_u32, c_u32)
        sig("ml_NewFP32", VP, VP, C.c_float)
        sig("ml_TensorData", c_f32p, VP)
        sig("ml_TensorShape", None, VP, c_u32p, c_u32p)
        sig("ml_TensorOp", C.c_int, VP)
        sig("ml_Nelements", c_u64, VP)
        sig("ml_TensorRead", C.c_int, VP, VP, c_f32p, c_u64)
        sig("ml_TensorDirty", None, VP)
        sig("ml_FreeTensor", None, VP)
        for name in ("ml_Add", "ml_Mul", "ml_MulMat", "ml_Repeat", "ml_GetRows", "ml_Copy", "ml_Scale"):
            sig(name, VP, VP, VP, VP)
        for name in ("ml_RMSNorm", "ml_SoftMax", "ml_Silu"):
            sig(name, VP, VP, VP)
        sig("ml_View1D", VP, VP, VP, c_u32, c_u32)
        sig("ml_Permute", VP, VP, VP, c_u32, c_u32, c_u32, c_u32)
        sig("ml_Rope", VP, VP, VP, c_u32, c_u32, c_u32)
        sig("ml_Reshape3D", VP, VP, VP, c_u32, c_u32, c_u32)
        sig("ml_DiagMaskInf", VP, VP, VP, c_u32)
        sig("ml_NewGraph", VP)
        sig("ml_FreeGraph", None, VP)
        sig("ml_BuildForwardExpand", C.c_int, VP, VP)
        sig("ml_GraphNodesCount", c_u32, VP)
        sig("ml_GraphNode", VP, VP, c_u32)
        sig("ml_GraphCompute", C.c_int, VP, VP)
        sig("llama_LoadModel", VP, C.c_char_p, c_u32)
        sig("llama_NewSyntheticModel", VP, C.POINTER(HParams), c_u64, c_u32, c_u32)
        sig("llama_SaveModel", C.c_int, VP, C.c_char_p, C.c_int)
        sig("llama_FreeModel", None, VP)
        sig("llama_ModelHParams", None, VP, C.POINTER(HParams))
        sig("llama_ModelFFSize", c_u32, VP)
        sig("llama_ModelTensor", VP, VP, C.c_char_p)
        sig("llama_NewContext", VP, VP, c_u32, C.c_int, C.c_int, C.c_int)
        sig("llama_ReleaseContext", None, VP)
        sig("llama_Eval", C.c_int, VP, VP, c_u32p, c_u32, c_u32)
        sig("llama_Logits", c_f32p, VP)
        sig("llama_MLContext", VP, VP)
        sig("llama_GreedyDecode", C.c_int, VP, VP, c_u32p, c_u32, c_u32, c_u32p, c_f32p)
        sig("llama_SampleTopPTopK", C.c_int, VP, c_f32p, c_u32, c_u32p, c_u32, c_u32, C.c_float, C.c_float, C.c_float, c_u64, c_u64, c_u32p)
        sig("llama_SampleDecode", C.c_int, VP, VP, c_u32p, c_u32, c_u32, c_u32, C.c_float, C.c_float, C.c_float, c_u64, c_u32p)
        sig("llamago_SampleDebug", C.c_int, VP, c_f32p, c_u32, c_u32p, c_u32, c_u32, C.c_float, C.c_float, C.c_float, c_u64, c_u64, c_u32p,
            c_u32p, c_f32p, c_u32p)

    # ---- helpers -------------------------------------------------------------------------------
    def last_error(self):
        return (self.lib.ml_LastError() or b"").decode()

    def _chk(self, handle, what):
        if not handle:
            raise MLError(f"{what}: {self.last_error()}")
        return handle

    # ---- llama.SampleTopPTopK (llama.go:455-707) ------------------------------------------------
    def SampleTopPTopK(self, ctx, logits, lastNTokens, topK=40, topP=0.95, temp=0.8, repeatPenalty=1.10, seed=0, draw=0, debug=False):
        """One sampling call.  debug=True also returns (candidate ids, probabilities) in rank order after the topP rescale."""
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        ring = (c_u32 * max(1, len(lastNTokens)))(*[int(t) for t in lastNTokens])
        tok = c_u32(0)
        if not debug:
            rc = self.lib.llama_SampleTopPTopK(ctx, lg.ctypes.data_as(c_f32p), lg.size, ring, len(lastNTokens), topK, topP, temp, repeatPenalty, seed, draw,
                                               C.byref(tok))
            if rc:
                raise MLError(f"llama_SampleTopPTopK: {self.last_error()}")
            return tok.value
        ids = (c_u32 * topK)()
        probs = np.zeros(topK, dtype=np.float32)
        keep = c_u32(0)
        rc = self.lib.llamago_SampleDebug(ctx, lg.ctypes.data_as(c_f32p), lg.size, ring, len(lastNTokens), topK, topP, temp, repeatPenalty, seed, draw,
                                          C.byref(tok), ids, probs.ctypes.data_as(c_f32p), C.byref(keep))
        if rc:
            raise MLError(f"llama_SampleTopPTopK: {self.last_error()}")
        return tok.value, list(ids)[:keep.value], probs[:keep.value].copy()

    # ---- ml.* ------------------------------------------------------------------------------------
    def NewContext(self, maxThreads=1, useAVX=False, useNEON=False):
        return self._chk(self.lib.ml_NewContext(maxThreads, int(useAVX), int(useNEON)), "ml_NewContext")

    def ReleaseContext(self, ctx):
        self.lib.ml_ReleaseContext(ctx)

    def NewTensor(self, ctx, ne, dt=TYPE_F32, data=None):
        ne = tuple(int(x) for x in ne)
        if len(ne) == 1:
            t = self.lib.ml_NewTensor1D(ctx, dt, ne[0])
        elif len(ne) == 2:
            t = self.lib.ml_NewTensor2D(ctx, dt, ne[0], ne[1])
        elif len(ne) == 3:
            t = self.lib.ml_NewTensor3D(ctx, dt, ne[0], ne[1], ne[2])
        else:
            raise ValueError("1..3 dims")
        self._chk(t, "ml_NewTensor")
        if data is not None:
            self.set_data(t, data)
        return t

    def NewFP32(self, ctx, v):
        return self._chk(self.lib.ml_NewFP32(ctx, float(v)), "ml_NewFP32")

    def data(self, t):
        """Host view of Tensor.Data as a flat numpy array (no copy)."""
        n = self.lib.ml_Nelements(t)
        return np.ctypeslib.as_array(self.lib.ml_TensorData(t), shape=(n,))

    def set_data(self, t, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32).ravel()
        d = self.data(t)
        assert a.size == d.size, (a.size, d.size)
        d[:] = a
        self.lib.ml_TensorDirty(t)

    def shape(self, t):
        ne = (c_u32 * 4)()
        nb = (c_u32 * 4)()
        self.lib.ml_TensorShape(t, ne, nb)
        return tuple(ne), tuple(nb)

    def read(self, ctx, t):
        """Computed values of a (contiguous) tensor, shaped numpy-style [ne3,ne2,ne1,ne0] squeezed of leading 1s."""
        n = self.lib.ml_Nelements(t)
        out = np.empty(n, dtype=np.float32)
        if self.lib.ml_TensorRead(ctx, t, out.ctypes.data_as(c_f32p), n):
            raise MLError(f"ml_TensorRead: {self.last_error()}")
        ne, _ = self.shape(t)
        return out.reshape(ne[3], ne[2], ne[1], ne[0])

    def op2(self, name, ctx, a, b):
        return self._chk(getattr(self.lib, "ml_" + name)(ctx, a, b), "ml_" + name)

    def op1(self, name, ctx, a):
        return self._chk(getattr(self.lib, "ml_" + name)(ctx, a), "ml_" + name)

    def Add(self, ctx, a, b): return self.op2("Add", ctx, a, b)
    def Mul(self, ctx, a, b): return self.op2("Mul", ctx, a, b)
    def MulMat(self, ctx, a, b): return self.op2("MulMat", ctx, a, b)
    def Repeat(self, ctx, a, b): return self.op2("Repeat", ctx, a, b)
    def GetRows(self, ctx, a, b): return self.op2("GetRows", ctx, a, b)
    def Copy(self, ctx, a, b): return self.op2("Copy", ctx, a, b)
    def Scale(self, ctx, a, b): return self.op2("Scale", ctx, a, b)
    def RMSNorm(self, ctx, a): return self.op1("RMSNorm", ctx, a)
    def SoftMax(self, ctx, a): return self.op1("SoftMax", ctx, a)
    def Silu(self, ctx, a): return self.op1("Silu", ctx, a)
    def View1D(self, ctx, a, ne0, off): return self._chk(self.lib.ml_View1D(ctx, a, ne0, off), "ml_View1D")
    def Permute(self, ctx, a, a0, a1, a2, a3): return self._chk(self.lib.ml_Permute(ctx, a, a0, a1, a2, a3), "ml_Permute")
    def Rope(self, ctx, a, past, dims, mode): return self._chk(self.lib.ml_Rope(ctx, a, past, dims, mode), "ml_Rope")
    def Reshape3D(self, ctx, a, n0, n1, n2): return self._chk(self.lib.ml_Reshape3D(ctx, a, n0, n1, n2), "ml_Reshape3D")
    def DiagMaskInf(self, ctx, a, past): return self._chk(self.lib.ml_DiagMaskInf(ctx, a, past), "ml_DiagMaskInf")

    def NewGraph(self):
        return self._chk(self.lib.ml_NewGraph(), "ml_NewGraph")

    def FreeGraph(self, g):
        self.lib.ml_FreeGraph(g)

    def BuildForwardExpand(self, g, t):
        if self.lib.ml_BuildForwardExpand(g, t):
            raise MLError(f"ml_BuildForwardExpand: {self.last_error()}")

    def GraphCompute(self, ctx, g):
        if self.lib.ml_GraphCompute(ctx, g):
            raise MLError(f"ml_GraphCompute: {self.last_error()}")

    def graph_ops(self, g):
        return [OP_NAMES[self.lib.ml_TensorOp(self.lib.ml_GraphNode(g, i))] for i in range(self.lib.ml_GraphNodesCount(g))]

    # ---- llama.* -----------------------------------------------------------------------------------
    def NewSyntheticModel(self, hp, seed=1234, layer0=0, layer1=0):
        return Model(self, self._chk(self.lib.llama_NewSyntheticModel(C.byref(hp), seed, layer0, layer1), "llama_NewSyntheticModel"))

    def LoadModel(self, path, ctxSize):
        return Model(self, self._chk(self.lib.llama_LoadModel(os.fsencode(path), ctxSize), "llama_LoadModel"))


def make_hparams(vocab, embd, mult, heads, layers, ctx=128):
    hp = HParams()
    hp.ctxSize, hp.vocabSize, hp.embdSize, hp.multSize, hp.headsCount, hp.layersCount = ctx, vocab, embd, mult, heads, layers
    hp.rotCount, hp.f16 = embd // heads, 0
    return hp


# Named shapes (SURVEY §8 / BASELINE.json configs).  "tiny" is the unit-test twin.
SHAPES = {
    "tiny": dict(vocab=512, embd=256, mult=256, heads=4, layers=2),
    "small": dict(vocab=2048, embd=1024, mult=256, heads=8, layers=4),
    "7B": dict(vocab=32000, embd=4096, mult=256, heads=32, layers=32),
    "13B": dict(vocab=32000, embd=5120, mult=256, heads=40, layers=40),
    "65B": dict(vocab=32000, embd=8192, mult=256, heads=64, layers=80),
}
PROMPT = [1, 306, 4658, 278, 6593, 310, 2834, 338]  # fixed 8-token prompt


class Model:
    """llama.Model (llama.go:181-193)."""

    def __init__(self, ml, handle):
        self.ml, self.h = ml, handle
        self.hp = HParams()
        ml.lib.llama_ModelHParams(handle, C.byref(self.hp))
        self.ffSize = ml.lib.llama_ModelFFSize(handle)

    def tensor(self, name):
        return self.ml.lib.llama_ModelTensor(self.h, name.encode())

    def Save(self, path, ftype=0):
        if self.ml.lib.llama_SaveModel(self.h, os.fsencode(path), ftype):
            raise MLError("llama_SaveModel failed")

    def QuantizeQ8(self):
        """Block-int8 weight matrices (our format): product quantises in HBM; a checker library replaces
        its weights by the dequantised values."""
        f = self.ml.lib.llamago_QuantizeModelQ8
        f.restype, f.argtypes = C.c_int, [VP]
        if f(self.h):
            raise MLError(f"llamago_QuantizeModelQ8: {self.ml.last_error()}")
        return self

    def NewContext(self, ctxSize=128, maxThreads=1, useAVX=False, useNEON=False):
        h = self.ml._chk(self.ml.lib.llama_NewContext(self.h, ctxSize, maxThreads, int(useAVX), int(useNEON)), "llama_NewContext")
        return Context(self, h)

    def free(self):
        if self.h:
            self.ml.lib.llama_FreeModel(self.h)
            self.h = None


class Context:
    """llama.Context (llama.go:83-88)."""

    def __init__(self, model, handle):
        self.model, self.ml, self.h = model, model.ml, handle

    def Eval(self, tokens, pastCount):
        toks = (c_u32 * len(tokens))(*[int(t) for t in tokens])
        if self.ml.lib.llama_Eval(self.h, self.model.h, toks, len(tokens), pastCount):
            raise MLError(f"llama_Eval: {self.ml.last_error()}")
        return self.logits()

    def logits(self):
        V = self.model.hp.vocabSize
        return np.ctypeslib.as_array(self.ml.lib.llama_Logits(self.h), shape=(V,)).copy()

    def GreedyDecode(self, prompt, n_predict, want_logits=True):
        V = self.model.hp.vocabSize
        toks = (c_u32 * len(prompt))(*[int(t) for t in prompt])
        out = (c_u32 * n_predict)()
        lg = np.empty((n_predict, V), dtype=np.float32) if want_logits else None
        rc = self.ml.lib.llama_GreedyDecode(self.h, self.model.h, toks, len(prompt), n_predict, out,
                                            lg.ctypes.data_as(c_f32p) if want_logits else None)
        if rc:
            raise MLError(f"llama_GreedyDecode: {self.ml.last_error()}")
        return list(out), lg

    def GreedyContinue(self, token, past, n_steps):
        """llamago_GreedyContinue: n_steps x { llama.Eval of one token through ml_GraphCompute, host argmax } from the context's present state."""
        L = self.ml.lib
        L.llamago_GreedyContinue.restype = C.c_int
        L.llamago_GreedyContinue.argtypes = [VP, VP, c_u32, c_u32, c_u32, c_u32p]
        out = (c_u32 * max(n_steps, 1))()
        if L.llamago_GreedyContinue(self.h, self.model.h, int(token), int(past), int(n_steps), out):
            raise MLError(f"llamago_GreedyContinue: {self.ml.last_error()}")
        return [int(t) for t in out[:n_steps]]

    def TimeComputes(self, on=True):
        """lh_ctx_time_computes: HIP events around every lh_graph_compute of this context; zeroes the sums."""
        L = self.ml.lib
        L.llamago_TimeComputes.restype = C.c_int
        L.llamago_TimeComputes.argtypes = [VP, C.c_int]
        if L.llamago_TimeComputes(self.h, 1 if on else 0):
            raise MLError(f"llamago_TimeComputes: {self.ml.last_error()}")

    def ComputeStats(self):
        """-> {"calls", "wall_us", "device_us"} summed over the lh_graph_compute calls since TimeComputes()."""
        L = self.ml.lib
        L.llamago_ComputeStats.restype = C.c_int
        L.llamago_ComputeStats.argtypes = [VP, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        n, w, d = C.c_uint64(0), C.c_double(0), C.c_double(0)
        if L.llamago_ComputeStats(self.h, C.byref(n), C.byref(w), C.byref(d)):
            raise MLError("llamago_ComputeStats failed")
        return {"calls": int(n.value), "wall_us": float(w.value), "device_us": float(d.value)}

    def EnableEmbedding(self):
        """ModelParams.Embedding (llama.go:52): every Eval also leaves row N-1 of `embeddings` in lctx.Embedding (llama.go:414-419)."""
        self.ml.lib.llamago_EnableEmbedding.restype = None
        self.ml.lib.llamago_EnableEmbedding.argtypes = [VP]
        self.ml.lib.llamago_EnableEmbedding(self.h)

    def Embedding(self):
        self.ml.lib.llama_Embedding.restype = c_f32p
        self.ml.lib.llama_Embedding.argtypes = [VP]
        p = self.ml.lib.llama_Embedding(self.h)
        return np.ctypeslib.as_array(p, shape=(self.model.hp.embdSize,)).copy() if p else None

    def SetKeepCount(self, keep):
        """ModelParams.KeepCount (llama.go:47): what a context swap keeps (server.go:166-167)."""
        self.ml.lib.llamago_SetKeepCount.restype = None
        self.ml.lib.llamago_SetKeepCount.argtypes = [VP, c_u32]
        self.ml.lib.llamago_SetKeepCount(self.h, keep)

    def SampleDecode(self, prompt, n_predict, topK=40, topP=0.95, temp=0.8, repeatPenalty=1.10, seed=0):
        """server.Do's generation loop (server.go:127-217) with SampleTopPTopK; returns the n_predict sampled ids."""
        toks = (c_u32 * len(prompt))(*[int(t) for t in prompt])
        out = (c_u32 * n_predict)()
        if self.ml.lib.llama_SampleDecode(self.h, self.model.h, toks, len(prompt), n_predict, topK, topP, temp, repeatPenalty, seed, out):
            raise MLError(f"llama_SampleDecode: {self.ml.last_error()}")
        return list(out)

    def free(self):
        if self.h:
            self.ml.lib.llama_ReleaseContext(self.h)
            self.h = None


def usable_threads():
    """Host threads a harness may use: affinity mask capped by the cgroup CPU quota (the GPU box shows 256 logical CPUs under a
    16-CPU quota; oversubscribing the checker's OpenMP loops there is catastrophic)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


class KernelTime(C.Structure):
    """lh_kernel_time (include/llamahip.h)."""
    _fields_ = [("name", C.c_char * 48), ("launches", c_u32), ("total_ms", C.c_float), ("bytes_per_launch", c_u64)]


def _bind_extensions(ml):
    """Product-only entry points (no counterpart in the reference): resident decode loop, kernel timing, pipeline stage."""
    L = ml.lib
    L.llamago_DecodeGreedyResident.restype = C.c_int
    L.llamago_DecodeGreedyResident.argtypes = [VP, c_u32, c_u32, c_u32, c_u32p, c_f32p]
    L.llamago_ProfileDecode.restype = C.c_int
    L.llamago_ProfileDecode.argtypes = [VP, c_u32, c_u32, c_u32, C.POINTER(KernelTime), c_u32]
    L.llamago_Stage.restype = C.c_int
    L.llamago_Stage.argtypes = [VP, c_u32p, VP, VP, VP, c_u32, c_u32, VP, VP]
    L.llamago_Sync.restype = C.c_int
    L.llamago_Sync.argtypes = [VP]
    L.llamago_SetStream.restype = None
    L.llamago_SetStream.argtypes = [VP]
    L.llamago_DeviceCount.restype = C.c_int
    L.llamago_HbmReadProbe.restype = C.c_int
    L.llamago_HbmReadProbe.argtypes = [c_u64, c_u32, c_f32p]
    L.llamago_LastGraphFused.restype = C.c_int
    L.llamago_LastGraphFused.argtypes = [VP]
    L.llamago_GraphComputeNoFusion.restype = C.c_int
    L.llamago_GraphComputeNoFusion.argtypes = [VP, VP]
    L.llamago_CommUniqueId.restype = C.c_int
    L.llamago_CommUniqueId.argtypes = [C.POINTER(C.c_uint8)]
    L.llamago_NewPipeline.restype = VP
    L.llamago_NewPipeline.argtypes = [VP, c_u32, c_u32, C.c_int, C.c_int, C.POINTER(C.c_uint8), VP]
    L.llamago_NewPipelineGrouped.restype = VP
    L.llamago_NewPipelineGrouped.argtypes = [VP, c_u32, c_u32, C.c_int, C.c_int, C.POINTER(C.c_uint8), VP, c_u32]
    L.llamago_PipelineGroups.restype = c_u32
    L.llamago_PipelineGroups.argtypes = [VP]
    L.llamago_PipelineRunSample.restype = C.c_int
    L.llamago_PipelineRunSample.argtypes = [VP, C.POINTER(c_u32p), c_u32p, c_u32, c_u32, C.c_float, C.c_float, C.c_float, c_u64, c_u32]
    L.llamago_NewBatch.restype = VP
    L.llamago_NewBatch.argtypes = [VP, c_u32, c_u32]
    L.llamago_FreeBatch.restype = None
    L.llamago_FreeBatch.argtypes = [VP]
    L.llamago_BatchBatched.restype = C.c_int
    L.llamago_BatchBatched.argtypes = [VP]
    L.llamago_BatchGreedyDecode.restype = C.c_int
    L.llamago_BatchGreedyDecode.argtypes = [VP, C.POINTER(c_u32p), c_u32p, c_u32, c_u32p, c_f32p]
    L.llamago_BatchPrompt.restype = C.c_int
    L.llamago_BatchPrompt.argtypes = [VP, C.POINTER(c_u32p), c_u32p, c_u32p]
    L.llamago_BatchTick.restype = C.c_int
    L.llamago_BatchTick.argtypes = [VP, c_u32p]
    L.llamago_FreePipeline.restype = None
    L.llamago_FreePipeline.argtypes = [VP]
    L.llamago_PipelineRun.restype = C.c_int
    L.llamago_PipelineRun.argtypes = [VP, C.POINTER(c_u32p), c_u32p, c_u32]
    L.llamago_PipelineTokens.restype = C.c_int
    L.llamago_PipelineTokens.argtypes = [VP, c_u32, c_u32p, c_u32]
    L.llamago_PipelineProfileDecode.restype = C.c_int
    L.llamago_PipelineProfileDecode.argtypes = [VP, c_u32, c_u32, c_u32, C.POINTER(KernelTime), c_u32]
    L.llamago_PipelineSync.restype = C.c_int
    L.llamago_PipelineSync.argtypes = [VP]
    ml.has_extensions = True


COMM_ID_BYTES = 128


def comm_unique_id(ml):
    """lh_comm_unique_id on this process's device: 128 bytes rank 0 hands to every other rank (any channel)."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    if ml.lib.llamago_CommUniqueId(buf):
        raise MLError(f"llamago_CommUniqueId: {ml.last_error()}")
    return bytes(buf)


class Pipeline:
    """Pods as pipeline streams over a layer-sharded model (server.go:84-106, 151): this rank's stages of `pods` independent
    greedy streams, scheduled below the C-ABI (lh_pipeline_run) with RCCL send/recv of the residual stream.
    comm_id: bytes from comm_unique_id (RCCL), or None with `hooks` (an lh_comm_hooks ctypes struct: host-staged
    transport), or both None for world == 1."""

    def __init__(self, model, ctxSize, pods, rank=0, world=1, comm_id=None, hooks=None, max_rows=0):
        """max_rows: streams one tick evaluates together in ONE pass over the rank's weights (0 = as many as fit, 1 = every stream
        on its own: one weight pass per stream and tick)."""
        self.model, self.ml, self.pods, self.rank, self.world, self.ctxSize = model, model.ml, pods, rank, world, ctxSize
        idbuf = (C.c_uint8 * COMM_ID_BYTES)(*comm_id) if comm_id is not None else None
        self._hooks = hooks  # keep the callbacks alive
        h = self.ml.lib.llamago_NewPipelineGrouped(model.h, ctxSize, pods, rank, world, idbuf, C.byref(hooks) if hooks is not None else None, max_rows)
        self.h = self.ml._chk(h, "llamago_NewPipeline")
        self.groups = int(self.ml.lib.llamago_PipelineGroups(self.h))

    def _prompt_args(self, prompts):
        assert len(prompts) == self.pods
        arrs = [(c_u32 * len(p))(*[int(t) for t in p]) for p in prompts]
        pp = (c_u32p * self.pods)(*[C.cast(a, c_u32p) for a in arrs])
        nn = (c_u32 * self.pods)(*[len(p) for p in prompts])
        return arrs, pp, nn

    def run_sample(self, prompts=None, steps=0, topK=40, topP=0.95, temp=0.8, repeatPenalty=1.10, seed=0, ringSize=0):
        """Like run(), with SampleTopPTopK on the last rank after every Eval (server.go:201-204); prompts on EVERY rank."""
        ring = ringSize or self.ctxSize
        if prompts is not None:
            keep, pp, nn = self._prompt_args(prompts)
            rc = self.ml.lib.llamago_PipelineRunSample(self.h, pp, nn, steps, topK, topP, temp, repeatPenalty, seed, ring)
        else:
            rc = self.ml.lib.llamago_PipelineRunSample(self.h, None, None, steps, topK, topP, temp, repeatPenalty, seed, ring)
        if rc:
            raise MLError(f"llamago_PipelineRunSample: {self.ml.last_error()}")

    def run(self, prompts=None, steps=0):
        """prompts: list of per-stream token lists (every rank passes them: the lengths shape the messages) or None to continue."""
        if prompts is not None:
            assert len(prompts) == self.pods
            arrs = [(c_u32 * len(p))(*[int(t) for t in p]) for p in prompts]
            pp = (c_u32p * self.pods)(*[C.cast(a, c_u32p) for a in arrs])
            nn = (c_u32 * self.pods)(*[len(p) for p in prompts])
            rc = self.ml.lib.llamago_PipelineRun(self.h, pp, nn, steps)
        else:
            rc = self.ml.lib.llamago_PipelineRun(self.h, None, None, steps)
        if rc:
            raise MLError(f"llamago_PipelineRun: {self.ml.last_error()}")

    def SetKeepCount(self, keep):
        """ModelParams.KeepCount of every stream (lh_pipeline_set_keep; the same value on every rank)."""
        self.ml.lib.llamago_PipelineSetKeepCount.argtypes = [VP, c_u32]
        if self.ml.lib.llamago_PipelineSetKeepCount(self.h, keep):
            raise MLError("llamago_PipelineSetKeepCount failed")

    def profile(self, on=True):
        """lh_pipeline_profile: HIP events around every stage and exchange of the following runs (clears the totals)."""
        self.ml.lib.llamago_PipelineProfile.argtypes = [VP, C.c_int]
        if self.ml.lib.llamago_PipelineProfile(self.h, 1 if on else 0):
            raise MLError("llamago_PipelineProfile failed")

    def stats(self):
        """Totals since profile(True): dict(ticks, stage_ms, exchange_ms) of THIS rank."""
        t, a, b = c_u32(0), C.c_float(0), C.c_float(0)
        self.ml.lib.llamago_PipelineStats.argtypes = [VP, c_u32p, c_f32p, c_f32p]
        if self.ml.lib.llamago_PipelineStats(self.h, C.byref(t), C.byref(a), C.byref(b)):
            raise MLError("llamago_PipelineStats failed")
        return dict(ticks=int(t.value), stage_ms=float(a.value), exchange_ms=float(b.value))

    def hop_probe(self, nbytes=16384, iters=1000):
        """Microseconds per ring shift of nbytes (every rank at the same time)."""
        us = C.c_float(0)
        self.ml.lib.llamago_PipelineHopProbe.argtypes = [VP, c_u32, c_u32, c_f32p]
        if self.ml.lib.llamago_PipelineHopProbe(self.h, nbytes, iters, C.byref(us)):
            raise MLError(f"llamago_PipelineHopProbe: {self.ml.last_error()}")
        return float(us.value)

    def tokens(self, pod):
        cap = 1 << 16
        out = (c_u32 * cap)()
        n = self.ml.lib.llamago_PipelineTokens(self.h, pod, out, cap)
        if n < 0:
            raise MLError(f"llamago_PipelineTokens: {self.ml.last_error()}")
        return list(out[:n])

    def profile_decode(self, token, past, repeats=2):
        arr = (KernelTime * 32)()
        n = self.ml.lib.llamago_PipelineProfileDecode(self.h, token, past, repeats, arr, 32)
        if n < 0:
            raise MLError(f"llamago_PipelineProfileDecode: {self.ml.last_error()}")
        return _kernel_times(arr, n)

    def free(self):
        if self.h:
            self.ml.lib.llamago_FreePipeline(self.h)
            self.h = None


class Batch:
    """`pods` llama.Contexts over one Model on ONE GPU whose decode steps share one pass over the weights (lh_batch; the reference
    runs its pods as independent goroutines, server.go:88-101)."""

    def __init__(self, model, ctxSize, pods):
        self.model, self.ml, self.pods = model, model.ml, pods
        self.h = self.ml._chk(self.ml.lib.llamago_NewBatch(model.h, ctxSize, pods), "llamago_NewBatch")
        self.batched = bool(self.ml.lib.llamago_BatchBatched(self.h))

    def GreedyDecode(self, prompts, n_predict, want_logits=False):
        """Every pod: its prompt as one Eval, then greedy steps; returns per pod the ids llama_GreedyDecode gives for its prompt alone."""
        assert len(prompts) == self.pods
        arrs = [(c_u32 * len(p))(*[int(t) for t in p]) for p in prompts]
        pp = (c_u32p * self.pods)(*[C.cast(a, c_u32p) for a in arrs])
        nn = (c_u32 * self.pods)(*[len(p) for p in prompts])
        out = (c_u32 * (self.pods * n_predict))()
        lg = np.empty((self.pods, self.model.hp.vocabSize), dtype=np.float32) if want_logits else None
        if self.ml.lib.llamago_BatchGreedyDecode(self.h, pp, nn, n_predict, out, lg.ctypes.data_as(c_f32p) if want_logits else None):
            raise MLError(f"llamago_BatchGreedyDecode: {self.ml.last_error()}")
        ids = [list(out[i * n_predict:(i + 1) * n_predict]) for i in range(self.pods)]
        return (ids, lg) if want_logits else ids

    def SetKeepCount(self, keep):
        self.ml.lib.llamago_BatchSetKeepCount.restype = None
        self.ml.lib.llamago_BatchSetKeepCount.argtypes = [VP, c_u32]
        self.ml.lib.llamago_BatchSetKeepCount(self.h, keep)

    def Prompt(self, prompts):
        """BatchHIP.Prompt (go/ml_hip_pods.go): every pod's prompt as one Eval; returns the id each prompt produced."""
        assert len(prompts) == self.pods
        arrs = [(c_u32 * len(p))(*[int(t) for t in p]) for p in prompts]
        pp = (c_u32p * self.pods)(*[C.cast(a, c_u32p) for a in arrs])
        nn = (c_u32 * self.pods)(*[len(p) for p in prompts])
        out = (c_u32 * self.pods)()
        if self.ml.lib.llamago_BatchPrompt(self.h, pp, nn, out):
            raise MLError(f"llamago_BatchPrompt: {self.ml.last_error()}")
        return list(out)

    def SetSampler(self, topK=40, topP=0.95, temp=0.8, repeatPenalty=1.10, seed=0, ringSize=64):
        """lh_batch_set_sampler: the following ticks sample (SampleTopPTopK on the device) instead of taking the argmax; allowed mid-stream."""
        f = self.ml.lib.llamago_BatchSetSampler
        f.restype, f.argtypes = C.c_int, [VP, c_u32, C.c_float, C.c_float, C.c_float, C.c_uint64, c_u32]
        if f(self.h, topK, topP, temp, repeatPenalty, seed, ringSize):
            raise MLError(f"llamago_BatchSetSampler: {self.ml.last_error()}")

    def Tick(self):
        """BatchHIP.Tick: one decode step of every pod in one pass over the weights; returns the ids produced."""
        out = (c_u32 * self.pods)()
        if self.ml.lib.llamago_BatchTick(self.h, out):
            raise MLError(f"llamago_BatchTick: {self.ml.last_error()}")
        return list(out)

    def free(self):
        if self.h:
            self.ml.lib.llamago_FreeBatch(self.h)
            self.h = None


def hbm_read_probe(ml, nbytes=4 << 30, repeats=4):
    """GB/s of a bare read-only HBM stream on this box (lh_hbm_read_probe)."""
    g = C.c_float(0)
    if ml.lib.llamago_HbmReadProbe(nbytes, repeats, C.byref(g)):
        raise MLError(f"llamago_HbmReadProbe: {ml.last_error()}")
    return float(g.value)


def decode_greedy_resident(ctx, first_token, past, n_steps, want_logits=False):
    """llamago_DecodeGreedyResident: n_steps decode steps without host round trips (argmax on the GPU)."""
    ml = ctx.ml
    out = (c_u32 * n_steps)()
    lg = np.empty(ctx.model.hp.vocabSize, dtype=np.float32) if want_logits else None
    if ml.lib.llamago_DecodeGreedyResident(ctx.h, first_token, past, n_steps, out, lg.ctypes.data_as(c_f32p) if want_logits else None):
        raise MLError(f"llamago_DecodeGreedyResident: {ml.last_error()}")
    return list(out), lg


def profile_decode(ctx, token, past, repeats=3):
    """Per-kernel-class HIP-event timing of eager decode steps: list of dicts(name, launches, avg_us, bytes_per_launch, gbps)."""
    ml = ctx.ml
    arr = (KernelTime * 32)()
    n = ml.lib.llamago_ProfileDecode(ctx.h, token, past, repeats, arr, 32)
    if n < 0:
        raise MLError(f"llamago_ProfileDecode: {ml.last_error()}")
    return _kernel_times(arr, n)


def _kernel_times(arr, n):
    res = []
    for i in range(n):
        k = arr[i]
        avg_us = k.total_ms * 1e3 / max(k.launches, 1)
        res.append(dict(name=k.name.decode(), launches=int(k.launches), avg_us=avg_us, bytes_per_launch=int(k.bytes_per_launch),
                        gbps=(k.bytes_per_launch / avg_us / 1e3) if avg_us > 0 else 0.0))
    return res


_product = None


def load_product():
    """The product library (HIP path).  Fails loudly when the extension is missing: there is no CPU fallback."""
    global _product
    if _product is None:
        from . import LIBLLAMAGO, LIBLLAMAHIP
        if not os.path.exists(LIBLLAMAHIP):
            raise MLError(f"HIP extension missing: {LIBLLAMAHIP} — build it with __graft_entry__.build(); no CPU fallback exists")
        hip = C.CDLL(LIBLLAMAHIP, mode=C.RTLD_GLOBAL)
        _product = MLLib(LIBLLAMAGO)
        _bind_extensions(_product)
        route_file = os.environ.get("LLAMAHIP_ROUTE_FILE")
        if route_file:   # test instrumentation (tests/test_gpu_zz_routes.py): this process appends the kernel families it launched when it exits
            import atexit
            hip.lh_route_log(1)
            hip.lh_route_names.restype = C.c_int64
            hip.lh_route_names.argtypes = [C.c_char_p, C.c_uint64]

            def _dump_routes():
                buf = C.create_string_buffer(int(hip.lh_route_names(None, 0)) + 16)
                hip.lh_route_names(buf, len(buf))
                with open(route_file, "a") as f:
                    f.write(buf.value.decode())
            atexit.register(_dump_routes)
    return _product
