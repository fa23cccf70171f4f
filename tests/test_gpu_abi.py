"""-m gpu: the raw C-ABI of include/llamahip.h driven from ctypes with hand-built lh_tensor arrays — exactly what the cgo shim
(llama.go_amd/go/ml_hip.go) does, without the C++ host mirror in between."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class LhTensor(C.Structure):  # struct lh_tensor (include/llamahip.h)
    _fields_ = [("op", C.c_uint8), ("dtype", C.c_uint8), ("flags", C.c_uint16), ("ne", C.c_uint32 * 4), ("nb", C.c_uint64 * 4),
                ("src0", C.c_int32), ("src1", C.c_int32), ("storage", C.c_int32), ("reserved", C.c_uint32), ("view_off", C.c_uint64),
                ("buf", C.c_uint64), ("host", C.POINTER(C.c_float))]


OP_NONE, OP_ADD, OP_MUL_MAT, OP_VIEW, OP_CPY, OP_SOFT_MAX = 0, 2, 20, 24, 22, 29


@pytest.fixture(scope="module")
def lh(product):
    import llama_go_amd as pkg
    lib = C.CDLL(pkg.LIBLLAMAHIP, mode=C.RTLD_GLOBAL)
    lib.lh_last_error.restype = C.c_char_p
    lib.lh_last_error.argtypes = [C.c_void_p]
    lib.lh_ctx_create.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.lh_ctx_destroy.argtypes = [C.c_void_p]
    lib.lh_tensor_register.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.lh_graph_compute.argtypes = [C.c_void_p, C.POINTER(LhTensor), C.c_uint32, C.c_uint32, C.c_uint32]
    lib.lh_node_read.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_float), C.c_uint64]
    lib.lh_buf_read.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(C.c_float), C.c_uint64]
    lib.lh_buf_free.argtypes = [C.c_void_p, C.c_uint64]
    return lib


def tensor(op, ne, src0=-1, src1=-1, storage=None, view_off=0, buf=0, host=None, index=None):
    t = LhTensor()
    t.op, t.dtype = op, 0
    ne = list(ne) + [1] * (4 - len(ne))
    for k in range(4):
        t.ne[k] = ne[k]
    t.nb[0] = 4
    for k in range(1, 4):
        t.nb[k] = t.nb[k - 1] * ne[k - 1]
    t.src0, t.src1 = src0, src1
    t.storage = index if storage is None else storage
    t.view_off, t.buf = view_off, buf
    if host is not None:
        t.host = host.ctypes.data_as(C.POINTER(C.c_float))
    return t


def test_struct_layout_matches_the_header(lh):
    assert C.sizeof(LhTensor) == 96  # 4 + 16 + 32 + 12 + 4 + 8 + 8 + 8, no padding surprises for the cgo side
    assert LhTensor.nb.offset == 24 and LhTensor.view_off.offset == 72 and LhTensor.host.offset == 88


def test_graph_compute_through_the_raw_abi(lh):
    ctx = C.c_void_p()
    assert lh.lh_ctx_create(0, None, C.byref(ctx)) == 0, lh.lh_last_error(None)
    rng = np.random.default_rng(3)
    K, M, N = 512, 300, 3
    w = (rng.standard_normal((M, K)) / np.sqrt(K)).astype(np.float32)
    x = rng.standard_normal((N, K)).astype(np.float32)
    b = rng.standard_normal((N, M)).astype(np.float32)
    # weight registered once (persistent, key = stable id), idempotent on the key
    ne = (C.c_uint32 * 4)(K, M, 1, 1)
    buf, buf2 = C.c_uint64(), C.c_uint64()
    assert lh.lh_tensor_register(ctx, 0xABCDEF, 0, ne, 1, w.ctypes.data, C.byref(buf)) == 0
    assert lh.lh_tensor_register(ctx, 0xABCDEF, 0, ne, 1, w.ctypes.data, C.byref(buf2)) == 0 and buf2.value == buf.value
    # leafs: 0 = W (persistent), 1 = x (host), 2 = bias (host); nodes: 3 = MulMat(W, x), 4 = Add(3, bias), 5 = SoftMax view of 4
    arr = (LhTensor * 6)()
    arr[0] = tensor(OP_NONE, (K, M), buf=buf.value, index=0)
    arr[1] = tensor(OP_NONE, (K, N), host=x, index=1)
    arr[2] = tensor(OP_NONE, (M, N), host=b, index=2)
    arr[3] = tensor(OP_MUL_MAT, (M, N), 0, 1, index=3)
    arr[4] = tensor(OP_ADD, (M, N), 3, 2, index=4)
    arr[5] = tensor(OP_SOFT_MAX, (M, N), 4, -1, storage=4, index=5)   # in-place op: a view of its source (ml.go:1005)
    assert lh.lh_graph_compute(ctx, arr, 3, 3, 0) == 0, lh.lh_last_error(ctx)
    out = np.empty((N, M), np.float32)
    assert lh.lh_node_read(ctx, 5, 0, out.ctypes.data_as(C.POINTER(C.c_float)), out.size) == 0
    z = x.astype(np.float64) @ w.astype(np.float64).T + b
    z = np.exp(z - z.max(axis=1, keepdims=True))
    z /= z.sum(axis=1, keepdims=True)
    assert np.abs(out - z).max() / z.max() < 1e-5
    # node 4 shares storage with node 5 (softmax wrote through)
    out4 = np.empty((N, M), np.float32)
    assert lh.lh_node_read(ctx, 4, 0, out4.ctypes.data_as(C.POINTER(C.c_float)), out4.size) == 0
    assert np.array_equal(out4, out)
    # errors are codes + messages, never aborts
    arr[3].src0 = 99
    assert lh.lh_graph_compute(ctx, arr, 3, 3, 0) < 0 and b"out of range" in lh.lh_last_error(ctx)
    arr[3].src0 = 0
    arr[4].op = 3  # OP_SUB: the reference HALTs on it (ml.go:1542-1545)
    assert lh.lh_graph_compute(ctx, arr, 3, 3, 0) == -4 and b"Please implement" in lh.lh_last_error(ctx)
    wb = np.empty(8, np.float32)
    assert lh.lh_buf_read(ctx, buf.value, 0, wb.ctypes.data_as(C.POINTER(C.c_float)), 8) == 0 and np.array_equal(wb, w.reshape(-1)[:8])
    assert lh.lh_buf_free(ctx, buf.value) == 0
    lh.lh_ctx_destroy(ctx)


def test_block_int8_registered_from_host_blocks(lh):
    """lh_tensor_register with dtype 7: the host hands the interchange format - blocks {float d; int8 q[32]} (kernels_q8.h) - and the library
    de-interleaves them on the device into its int8 plane + scale plane (k_q8_deinterleave); read back dequantised they must equal fl32(d * q)
    of every block, bit for bit."""
    ctx = C.c_void_p()
    assert lh.lh_ctx_create(0, None, C.byref(ctx)) == 0, lh.lh_last_error(None)
    rng = np.random.default_rng(11)
    K, M = 352, 77                                   # 11 blocks per row
    w = (rng.standard_normal((M, K)) * 0.05).astype(np.float32)
    wb = w.reshape(-1, 32)
    d = (np.abs(wb).max(axis=1) / np.float32(127.0)).astype(np.float32)
    q = np.clip(np.rint(wb / np.where(d > 0, d, 1)[:, None]), -127, 127).astype(np.int8)
    blocks = np.zeros(len(d), dtype=np.dtype([("d", "<f4"), ("q", "i1", 32)]))
    blocks["d"], blocks["q"] = d, q
    assert blocks.itemsize == 36
    ne = (C.c_uint32 * 4)(K, M, 1, 1)
    buf = C.c_uint64()
    assert lh.lh_tensor_register(ctx, 0, 7, ne, 1, blocks.ctypes.data, C.byref(buf)) == 0, lh.lh_last_error(ctx)
    back = np.empty(M * K, np.float32)
    assert lh.lh_buf_read(ctx, buf.value, 0, back.ctypes.data_as(C.POINTER(C.c_float)), back.size) == 0, lh.lh_last_error(ctx)
    want = (d[:, None] * q.astype(np.float32)).astype(np.float32).reshape(-1)
    assert np.array_equal(back, want)
    part = np.empty(100, np.float32)                 # an unaligned range crossing blocks
    assert lh.lh_buf_read(ctx, buf.value, 45, part.ctypes.data_as(C.POINTER(C.c_float)), part.size) == 0
    assert np.array_equal(part, want[45:145])
    assert lh.lh_buf_free(ctx, buf.value) == 0
    lh.lh_ctx_destroy(ctx)
