"""not-gpu: the checker (oracle/oracle.c) and the product's host mirror (llama.go_amd/host/llamago.cpp) restate the same Go graph
builders twice (ml.go:241-1043, llama.go:232-387).  A shared drift would fake a parity pass, a one-sided drift would break the
fused-plan matcher: both libraries export llamago_DescribeEvalGraph (the Eval graph as numbers, no GPU needed) and the two
descriptions must be identical tensor by tensor — op, NE, NB, source indices, leaf/node order — for decode, the 8-token prompt and
a 40-token prompt, at past = 0 and past > 0."""
import ctypes as C
import os

import numpy as np
import pytest

from llama_go_amd.mlapi import SHAPES, make_hparams, HParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def describe(lib, hp, ctx, N, past):
    f = lib.llamago_DescribeEvalGraph
    f.restype = C.c_int
    f.argtypes = [C.POINTER(HParams), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_uint32)]
    nl = C.c_uint32(0)
    n = f(C.byref(hp), ctx, N, past, None, 0, C.byref(nl))
    assert n > 0
    buf = np.zeros((n, 11), dtype=np.int32)
    assert f(C.byref(hp), ctx, N, past, buf.ctypes.data_as(C.POINTER(C.c_int32)), n, C.byref(nl)) == n
    return buf, nl.value


@pytest.mark.parametrize("shape", ["tiny", "small"])
@pytest.mark.parametrize("N,past", [(1, 0), (1, 23), (8, 0), (40, 0), (40, 17)])
def test_checker_and_host_mirror_build_the_same_graph(built, shape, N, past):
    import llama_go_amd as pkg
    C.CDLL(pkg.LIBLLAMAHIP, mode=C.RTLD_GLOBAL)
    prod = C.CDLL(pkg.LIBLLAMAGO)
    orc = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
    hp = make_hparams(**SHAPES[shape], ctx=64)
    a, nla = describe(prod, hp, 64, N, past)
    b, nlb = describe(orc, hp, 64, N, past)
    assert nla == nlb and a.shape == b.shape
    bad = np.nonzero((a != b).any(axis=1))[0]
    assert bad.size == 0, f"first differing tensor {bad[0]}: host mirror {a[bad[0]].tolist()} vs checker {b[bad[0]].tolist()}"
    L = hp.layersCount
    nodes = a[nla:]
    # llama.go's per-layer op sequence: 2 Repeat nodes per layer and one after the last layer exist only when N > 1
    # (37 compute+view nodes per layer at N = 1, 39 otherwise) + GetRows, final RMSNorm [+ Repeat], Mul, lm_head
    assert len(nodes) == (L * 37 + 4 if N == 1 else L * 39 + 5), len(nodes)
    ops = nodes[:, 0]
    assert (ops == 20).sum() == 9 * L + 1            # MUL_MAT: 7 weight + KQ + KQV per layer, + lm_head
    assert (ops == 27).sum() == 1 and ops[-1] == 20  # one GET_ROWS; the graph ends in the lm_head
    assert (ops == 10).sum() == (0 if N == 1 else 2 * L + 1)  # REPEAT only when N > 1 (ml.go:496-498)


def test_two_graphs_on_one_thread_do_not_free_each_other(built):
    """ml_FreeGraph releases what ITS graph reached, not every tensor the thread made since the last FreeGraph."""
    from llama_go_amd.mlapi import load_product, MLLib
    for ml in (load_product(), MLLib(os.path.join(ROOT, "oracle", "liboracle.so"))):
        a = ml.NewTensor(None, (8, 4))
        b = ml.NewTensor(None, (8, 4))
        g1, g2 = ml.NewGraph(), ml.NewGraph()
        n1 = ml.SoftMax(None, ml.Scale(None, ml.Add(None, a, b), ml.NewFP32(None, 0.5)))
        ml.BuildForwardExpand(g1, n1)
        n2 = ml.Silu(None, ml.Mul(None, a, b))
        ml.BuildForwardExpand(g2, n2)
        ml.FreeGraph(g1)
        assert ml.graph_ops(g2) == ["MUL", "SILU"]       # g2's nodes are still alive and intact
        ne, nb = ml.shape(n2)
        assert ne == (8, 4, 1, 1) and nb[0] == 4
        ml.FreeGraph(g2)


def describe_array(lib, hp, ctx, N, past_build, past_query):
    f = lib.llamago_DescribeEvalArray
    f.restype = C.c_int
    f.argtypes = [C.POINTER(HParams), C.c_uint32, C.c_uint32, C.c_int64, C.c_uint32, C.POINTER(C.c_int64), C.c_uint32, C.POINTER(C.c_uint32)]
    nl = C.c_uint32(0)
    n = f(C.byref(hp), ctx, N, past_build, past_query, None, 0, C.byref(nl))
    assert n > 0, n
    buf = np.zeros((n, 16), dtype=np.int64)
    assert f(C.byref(hp), ctx, N, past_build, past_query, buf.ctypes.data_as(C.POINTER(C.c_int64)), n, C.byref(nl)) == n
    return buf, nl.value


@pytest.mark.parametrize("shape", ["tiny", "small"])
@pytest.mark.parametrize("N", [1, 3])
def test_kept_decode_graph_moved_to_a_position_equals_a_fresh_build(built, shape, N):
    """llama_Eval keeps the graph of a one-token Eval between calls and MOVES it (host/llamago.cpp, eval_cache: every number of the flattened
    array is affine in `past`; the differences are learnt from three builds, nothing about shapes is restated).  The array it then hands to
    lh_graph_compute must be the one a fresh build at that position produces: every field of every tensor, and the Rope / DiagMaskInf
    parameter values and token ids (folded), over every position of the window from learning points at its start, middle and end."""
    import llama_go_amd as pkg
    C.CDLL(pkg.LIBLLAMAHIP, mode=C.RTLD_GLOBAL)
    prod = C.CDLL(pkg.LIBLLAMAGO)
    ctx = 48
    hp = make_hparams(**SHAPES[shape], ctx=ctx)
    fresh = {p: describe_array(prod, hp, ctx, N, -1, p) for p in range(0, ctx - N + 1)}
    moved_fields = set()
    for pb in (0, 17, ctx - N):                      # (the last one: learnt from the three positions that still fit below it)
        for pq in range(0, ctx - N + 1):
            a, nla = describe_array(prod, hp, ctx, N, pb, pq)
            b, nlb = fresh[pq]
            assert nla == nlb and a.shape == b.shape
            bad = np.nonzero((a != b).any(axis=1))[0]
            assert bad.size == 0, f"learnt at {pb}, moved to {pq}: tensor {bad[0]} kept {a[bad[0]].tolist()} vs fresh {b[bad[0]].tolist()}"
        moved_fields |= set(np.nonzero((fresh[0][0] != fresh[5][0]).any(axis=0))[0].tolist())
    # what moves with the position: extents (3..6), strides (7..10), view offsets (14), parameter values (15) - never structure
    assert moved_fields and moved_fields <= {3, 4, 5, 6, 7, 8, 9, 10, 14, 15}, moved_fields


def test_kept_decode_graph_declines_a_window_it_cannot_learn_in(built):
    import llama_go_amd as pkg
    C.CDLL(pkg.LIBLLAMAHIP, mode=C.RTLD_GLOBAL)
    prod = C.CDLL(pkg.LIBLLAMAGO)
    hp = make_hparams(**SHAPES["tiny"], ctx=2)
    f = prod.llamago_DescribeEvalArray
    f.restype = C.c_int
    f.argtypes = [C.POINTER(HParams), C.c_uint32, C.c_uint32, C.c_int64, C.c_uint32, C.POINTER(C.c_int64), C.c_uint32, C.POINTER(C.c_uint32)]
    assert f(C.byref(hp), 2, 1, 0, 0, None, 0, None) == -2     # three consecutive positions do not fit a 2-token window: Eval builds as before
    assert f(C.byref(hp), 2, 1, -1, 1, None, 0, None) > 0
