"""-m gpu: per-operator parity of the HIP kernels (node-by-node path of lh_graph_compute) against the oracle.

Each test builds the same small graph with the reference's operator constructors on both libraries
(product = MI355X through the C-ABI, oracle = CPU restatement) on the same seeded inputs and compares
the results.  Shapes cover the hot-path shapes of SURVEY §8a and its edge cases: past = 0, N = 1,
T < 8 (the reference's AVX defect zone), non-multiple-of-wave sizes.

Tolerances (written here as the contract):
  * element-wise / copy / mask / gather ops: bit-exact (same fp32 operation sequence);
  * reductions (MulMat, RMSNorm, SoftMax): the GPU sums in a tree, the reference left-to-right —
    max |delta| <= 1e-5 * max |ref| for dot products of <= 11008 terms, 2e-6 for the others.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rng(seed):
    return np.random.default_rng(seed)


class Pair:
    """Run the same graph-building function on both libraries."""

    def __init__(self, product, oracle):
        self.libs = {"hip": product, "orc": oracle}
        product.lib.llamago_GraphComputeNoFusion.restype = C.c_int
        product.lib.llamago_GraphComputeNoFusion.argtypes = [C.c_void_p, C.c_void_p]

    def run(self, build, n_out=1):
        """build(ml, ctx) -> list of result tensors (last one is the graph root).  Returns {name: [arrays]}."""
        out = {}
        for name, ml in self.libs.items():
            ctx = ml.NewContext(4, False, False)
            g = ml.NewGraph()
            res = build(ml, ctx)
            if not isinstance(res, (list, tuple)):
                res = [res]
            for t in res:
                ml.BuildForwardExpand(g, t)
            if name == "hip":
                if ml.lib.llamago_GraphComputeNoFusion(ctx, g):
                    raise RuntimeError(ml.last_error())
            else:
                ml.GraphCompute(ctx, g)
            out[name] = [ml.read(ctx, t).copy() for t in res]
            ml.FreeGraph(g)
            ml.ReleaseContext(ctx)
        return out


@pytest.fixture(scope="module")
def pair(product, oracle):
    return Pair(product, oracle)


def leaf(ml, ctx, arr):
    """numpy array [.., ne1, ne0] -> leaf tensor with NE = reversed shape."""
    arr = np.asarray(arr, dtype=np.float32)
    return ml.NewTensor(ctx, tuple(reversed(arr.shape)), data=arr)


def close(a, b, tol):
    denom = max(float(np.abs(b).max()), 1e-30)
    return float(np.abs(a - b).max()) / denom <= tol


def assert_exact(out):
    for a, b in zip(out["hip"], out["orc"]):
        assert a.shape == b.shape
        assert np.array_equal(a, b, equal_nan=True), f"max abs diff {np.abs(a - b).max()}"


def assert_close(out, tol):
    for a, b in zip(out["hip"], out["orc"]):
        assert a.shape == b.shape
        assert close(a, b, tol), f"rel err {np.abs(a - b).max() / max(np.abs(b).max(), 1e-30):.3e} > {tol}"


@pytest.mark.parametrize("K,M,N", [(4096, 512, 1), (11008, 256, 1), (4096, 300, 3), (128, 7, 5), (24, 128, 1), (100, 33, 2), (4, 2, 1),
                                   (4096, 512, 64), (1024, 384, 130), (11008, 256, 33),
                                   # N >= 32: MFMA GEMMs.  K >= 512 -> persistent LDS-DMA kernel (ragged M and N, K = 512 edge,
                                   # more tiles than CUs); smaller K -> the register-staged kernel
                                   (2048, 300, 200), (5120, 700, 257), (512, 256, 32), (480, 256, 40), (1024, 4200, 1100),
                                   # long prompts x wide matrices: the exact bf16 x 9 GEMM (k_gemm_b9; ragged M and N, K = 512 edge)
                                   (1024, 5000, 1100), (512, 5200, 1030), (2048, 10496, 513)])
def test_mul_mat_weights(pair, K, M, N):
    r = rng(K + M + N)
    w = r.standard_normal((M, K)).astype(np.float32) / np.sqrt(K)
    x = r.standard_normal((N, K)).astype(np.float32)
    out = pair.run(lambda ml, ctx: ml.MulMat(ctx, leaf(ml, ctx, w), leaf(ml, ctx, x)))
    assert out["hip"][0].shape == (1, 1, N, M)
    assert_close(out, 1e-5)
    ref64 = (x.astype(np.float64) @ w.astype(np.float64).T).reshape(1, 1, N, M)
    assert close(out["hip"][0], ref64, 1e-5)


@pytest.mark.parametrize("N", [1, 5, 8, 33, 128, 300])
@pytest.mark.parametrize("K,M", [(4096, 11008), (11008, 4096), (4096, 4096)])
def test_mul_mat_full_7b_matrices_exact_properties(product, K, M, N):
    """BASELINE's full matrix sizes (LLaMA-7B: w1 / w3 11008 x 4096, w2 4096 x 11008, wq..wo 4096 x 4096) cost the checker's scalar loop minutes per
    product, so at these sizes the MulMat of the path (ml.go:1976-2098) is held to three properties that need no checker, are independent of the
    size, and hold BIT FOR BIT whatever order a kernel sums in - as long as that order depends on nothing but the contraction index:
      (1) powers of two commute with every rounding:  MulMat(4 W, x / 2) == 2 MulMat(W, x)
      (2) a weight row's result does not depend on where the row stands:  MulMat(W[p], x) == MulMat(W, x)[:, p]
      (3) a token row's result does not depend on its neighbours:  MulMat(W, x[q]) == MulMat(W, x)[q]
    and the values themselves to a float64 product within the tolerance of the small cases above."""
    ml = product
    ml.lib.llamago_GraphComputeNoFusion.restype = C.c_int
    ml.lib.llamago_GraphComputeNoFusion.argtypes = [C.c_void_p, C.c_void_p]
    r = rng(K * 7 + M * 3 + N)
    w = (r.standard_normal((M, K), dtype=np.float32) / np.float32(np.sqrt(K))).astype(np.float32)
    x = r.standard_normal((N, K), dtype=np.float32)
    p, q = r.permutation(M), r.permutation(N)
    ctx = ml.NewContext(1, False, False)
    g = ml.NewGraph()
    try:
        W, X = leaf(ml, ctx, w), leaf(ml, ctx, x)
        ys = [ml.MulMat(ctx, W, X),
              ml.MulMat(ctx, leaf(ml, ctx, w * np.float32(4)), leaf(ml, ctx, x * np.float32(0.5))),
              ml.MulMat(ctx, leaf(ml, ctx, w[p]), X),
              ml.MulMat(ctx, W, leaf(ml, ctx, x[q]))]
        for t in ys:
            ml.BuildForwardExpand(g, t)
        if ml.lib.llamago_GraphComputeNoFusion(ctx, g):
            raise RuntimeError(ml.last_error())
        y, y_scaled, y_rows, y_tokens = (ml.read(ctx, t).reshape(N, M).copy() for t in ys)
    finally:
        ml.FreeGraph(g)
        ml.ReleaseContext(ctx)
    assert np.array_equal(y_scaled, y * np.float32(2)), f"scaling by powers of two changed {np.count_nonzero(y_scaled != y * np.float32(2))} sums"
    assert np.array_equal(y_rows, y[:, p]), f"{np.count_nonzero(y_rows != y[:, p])} sums depend on the weight row's place"
    assert np.array_equal(y_tokens, y[q]), f"{np.count_nonzero(y_tokens != y[q])} sums depend on the token row's place"
    ref64 = x.astype(np.float64) @ w.astype(np.float64).T
    assert close(y, ref64, 1e-5)


@pytest.mark.parametrize("T,N,H,hd,past", [(1, 1, 4, 64, 0), (7, 1, 4, 128, 6), (24, 8, 2, 128, 16), (5, 5, 3, 32, 0)])
def test_attention_chain(pair, T, N, H, hd, past):
    """KQ (strided operands) -> Scale -> DiagMaskInf -> SoftMax -> KQV through a transposing Cpy -> merge Cpy
    (llama.go:281-333), with T < 8 included."""
    assert T == past + N
    d = H * hd
    r = rng(T * 31 + N)
    kc = r.standard_normal((T, d)).astype(np.float32)
    vc = r.standard_normal((T, d)).astype(np.float32)
    q = r.standard_normal((N, d)).astype(np.float32)

    def build(ml, ctx):
        K3 = ml.Permute(ctx, ml.Reshape3D(ctx, leaf(ml, ctx, kc.reshape(-1)), hd, H, T), 0, 2, 1, 3)
        Q3 = ml.Permute(ctx, ml.Copy(ctx, leaf(ml, ctx, q), ml.NewTensor(ctx, (hd, H, N))), 0, 2, 1, 3)
        KQ = ml.MulMat(ctx, K3, Q3)
        S = ml.SoftMax(ctx, ml.DiagMaskInf(ctx, ml.Scale(ctx, KQ, ml.NewFP32(ctx, 1.0 / np.sqrt(hd))), past))
        VT = ml.Copy(ctx, ml.Permute(ctx, ml.Reshape3D(ctx, leaf(ml, ctx, vc.reshape(-1)), hd, H, T), 1, 2, 0, 3), ml.NewTensor(ctx, (T, hd, H)))
        KQV = ml.MulMat(ctx, VT, S)
        merged = ml.Copy(ctx, ml.Permute(ctx, KQV, 0, 2, 1, 3), ml.NewTensor(ctx, (d, N)))
        return [S, merged]

    out = pair.run(build)
    assert_close(out, 2e-6)
    # masked probabilities are exactly zero on both sides
    S = out["hip"][0][0]
    for j in range(N):
        assert np.all(S[:, j, past + j + 1:] == 0.0)


@pytest.mark.parametrize("d,N", [(4096, 1), (4096, 3), (256, 2), (100, 1)])
def test_rms_norm_repeat_mul(pair, d, N):
    r = rng(d + N)
    x = (r.standard_normal((N, d)) * 3).astype(np.float32)
    g = (1 + 0.1 * r.standard_normal(d)).astype(np.float32)

    def build(ml, ctx):
        cur = ml.RMSNorm(ctx, leaf(ml, ctx, x))
        return ml.Mul(ctx, ml.Repeat(ctx, leaf(ml, ctx, g), cur), cur)

    out = pair.run(build)
    assert_close(out, 2e-6)


@pytest.mark.parametrize("mode,past,N", [(0, 0, 1), (0, 5, 3), (1, 0, 4), (1, 6, 2), (0, 100, 1)])
def test_rope(pair, mode, past, N):
    hd, H = 128, 3
    n2 = N if mode == 0 else past + N
    r = rng(mode * 100 + past + N)
    x = r.standard_normal((n2, H, hd)).astype(np.float32)
    out = pair.run(lambda ml, ctx: ml.Rope(ctx, leaf(ml, ctx, x), past, hd, mode))
    # f64 table on both sides, rotation in f64, one rounding: bit-exact
    assert_exact(out)
    if mode == 1 and past > 0:  # rows before `past` untouched (cache keeps already-rotated keys)
        assert np.array_equal(out["hip"][0][0, :past], x[:past])


def test_silu_add_scale_exact(pair):
    r = rng(7)
    a = (r.standard_normal((3, 1000)) * 4).astype(np.float32)
    b = r.standard_normal((3, 1000)).astype(np.float32)
    out = pair.run(lambda ml, ctx: ml.Add(ctx, ml.Mul(ctx, ml.Silu(ctx, leaf(ml, ctx, a)), leaf(ml, ctx, b)), leaf(ml, ctx, b)))
    # exp() comes from two different libm implementations (ocml / glibc), both <= 1 ulp in f64: the fp32 results
    # can differ in the last bit on rare elements
    assert_close(out, 2e-7)


def test_get_rows_and_copy_exact(pair):
    r = rng(11)
    emb = r.standard_normal((50, 64)).astype(np.float32)
    ids = np.array([3, 49, 0, 3], dtype=np.float32)
    out = pair.run(lambda ml, ctx: ml.GetRows(ctx, leaf(ml, ctx, emb), leaf(ml, ctx, ids)))
    assert_exact(out)
    assert np.array_equal(out["hip"][0][0, 0], emb[[3, 49, 0, 3]])
    x = r.standard_normal((4, 6, 8)).astype(np.float32)
    for perm in [(1, 2, 0, 3), (0, 2, 1, 3), (2, 0, 1, 3)]:
        def build(ml, ctx, perm=perm):
            p = ml.Permute(ctx, leaf(ml, ctx, x), *perm)
            ne, _ = ml.shape(p)
            return ml.Copy(ctx, p, ml.NewTensor(ctx, ne[:3]))
        assert_exact(pair.run(build))


def test_view_copy_into_cache(pair):
    """Cpy into a View1D slot of a larger buffer, then read the whole buffer back (llama.go:274-278)."""
    r = rng(13)
    cache = np.zeros(5 * 16, dtype=np.float32)
    cur = r.standard_normal((2, 16)).astype(np.float32)

    def build(ml, ctx):
        c = leaf(ml, ctx, cache)
        st = ml.Copy(ctx, leaf(ml, ctx, cur), ml.View1D(ctx, c, 32, 16 * 2))
        whole = ml.Copy(ctx, ml.View1D(ctx, c, 80, 0), ml.NewTensor(ctx, (80,)))
        return [st, whole]

    out = pair.run(build)
    assert_exact(out)
    want = cache.copy()
    want[32:64] = cur.reshape(-1)
    assert np.array_equal(out["hip"][1].reshape(-1), want)


def test_unimplemented_op_reports_halt(product):
    """Ops the reference HALTs on must come back as an error code + message, never a crash (ml.go:1536-1700)."""
    ml = product
    ctx = ml.NewContext(1)
    a = ml.NewTensor(ctx, (4,), data=np.ones(4, np.float32))
    with pytest.raises(Exception):
        ml.Mul(ctx, a, ml.NewTensor(ctx, (5,)))  # shape mismatch: "[STOP] MulImpl"
    assert "MulImpl" in ml.last_error()
    ml.ReleaseContext(ctx)
