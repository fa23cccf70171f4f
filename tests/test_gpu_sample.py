"""GPU parity of the device sampler (csrc/kernels_sample.h) against the checker's SampleTopPTopK / SampleDecode
(llama.go:455-707, server.go:127-217), through libllamago.so -> libllamahip.so."""
import os

import numpy as np
import pytest

from llama_go_amd.mlapi import SHAPES, MLError, make_hparams

pytestmark = pytest.mark.gpu


def _logits(rng, V, kind):
    x = rng.standard_normal(V).astype(np.float32) * 4
    if kind == "ties":
        x = np.round(x * 2) / 2
    if kind == "neginf":
        x[rng.integers(0, V, V // 3)] = -np.inf
    if kind == "flat":
        x[:] = -0.75
    if kind == "zeros":        # +0 / -0 compare equal in the reference: ids decide
        x[: V // 2] = 0.0
        x[1: V // 2: 2] = -0.0
        x[V // 2:] = -1.0
    return x.astype(np.float32)


def _ulp_close(a, b, ulps=4):
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    return np.all(np.abs(a.astype(np.float64) - b.astype(np.float64)) <= ulps * np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)))


@pytest.mark.parametrize("kind", ["normal", "ties", "neginf", "flat", "zeros"])
@pytest.mark.parametrize("V,topK,topP", [(512, 40, 0.95), (2048, 1, 0.95), (2048, 100, 0.5), (32000, 40, 0.95), (32000, 40, 1.0), (32000, 1024, 0.9),
                                         (50000, 64, 0.95), (1000, 1000, 1.0)])
def test_sampler_matches_checker(product, oracle, kind, V, topK, topP):
    rng = np.random.default_rng(V * 7 + topK)
    hctx = product.NewContext(1)
    octx = oracle.NewContext(1)
    for draw in range(3):
        lg = _logits(rng, V, kind)
        ring = [0] * 8 + [int(t) for t in rng.integers(0, V, 120)]
        tok, ids, probs = product.SampleTopPTopK(hctx, lg, ring, topK, topP, 0.8, 1.1, seed=2024, draw=draw, debug=True)
        otok, oids, oprobs = oracle.SampleTopPTopK(octx, lg, ring, topK, topP, 0.8, 1.1, seed=2024, draw=draw, debug=True)
        assert ids == oids                      # integers: exact (selection, tie order, topP cut)
        assert _ulp_close(probs, oprobs)        # device exp vs libm exp in f64, then the same fp32 steps
        assert tok == otok
        assert product.SampleTopPTopK(hctx, lg, ring, topK, topP, 0.8, 1.1, seed=2024, draw=draw) == tok


def test_sampler_distribution_follows_the_reference_rule(product):
    """Size-independent property at the full vocabulary: over many draws the pick frequencies follow the reference's own
    'argmax p^2 f^2' rule (llama.go:661-673), which for two candidates with p1 > p2 picks the first with probability 1 - p2/(2 p1)."""
    hctx = product.NewContext(1)
    V = 32000
    lg = np.full(V, -30.0, np.float32)
    lg[123], lg[4567] = 1.0, 0.0                         # temp 1: p = (e, 1) / (e + 1)
    n = 600
    picks = [product.SampleTopPTopK(hctx, lg, [], 2, 1.0, 1.0, 1.0, seed=5, draw=d) for d in range(n)]
    assert set(picks) <= {123, 4567}
    p1, p2 = np.e / (np.e + 1), 1 / (np.e + 1)
    want = 1 - p2 / (2 * p1)
    got = picks.count(123) / n
    assert abs(got - want) < 4 * np.sqrt(want * (1 - want) / n)


def test_sampler_errors(product):
    hctx = product.NewContext(1)
    lg = np.zeros(1000, np.float32)
    for bad, msg in ((dict(topK=0), "topK"), (dict(topK=1001), "topK"), (dict(temp=0.0), "temp"), (dict(repeatPenalty=0.0), "repeatPenalty")):
        kw = dict(topK=40, topP=0.95, temp=0.8, repeatPenalty=1.1)
        kw.update(bad)
        with pytest.raises(MLError, match=msg):
            product.SampleTopPTopK(hctx, lg, [], seed=1, **kw)
    with pytest.raises(MLError, match="device limit"):
        product.SampleTopPTopK(hctx, np.zeros(4000, np.float32), [], topK=2000, seed=1)
    with pytest.raises(MLError, match="vocabulary"):
        product.SampleTopPTopK(hctx, np.zeros(70000, np.float32), [], topK=40, seed=1)


def _sample_decode(lib, shape, ctx, prompt, n_predict, seed, **kw):
    hp = make_hparams(**SHAPES[shape], ctx=ctx)
    m = lib.NewSyntheticModel(hp, 1234)
    c = m.NewContext(ctx, 16, False)
    out = c.SampleDecode(prompt, n_predict, seed=seed, **kw)
    c.free()
    m.free()
    return out


@pytest.mark.parametrize("shape,prompt,ctx", [("tiny", [1, 5, 9, 200, 17, 3, 44, 100], 64), ("tiny", [7], 32),
                                              ("small", [1, 306, 1658, 278, 1593, 310, 834, 338], 64)])
@pytest.mark.parametrize("graph", [True, False])
def test_sample_decode_matches_checker(product, oracle, shape, prompt, ctx, graph, monkeypatch):
    if not graph:
        monkeypatch.setenv("LLAMAHIP_NO_GRAPH", "1")
    kw = dict(topK=40, topP=0.95, temp=0.8, repeatPenalty=1.1)
    n = ctx - len(prompt) + 1                      # runs to the last position of the context window
    got = _sample_decode(product, shape, ctx, prompt, n, 99, **kw)
    want = _sample_decode(oracle, shape, ctx, prompt, n, 99, **kw)
    assert got == want
    assert _sample_decode(product, shape, ctx, prompt, n, 100, **kw) != got
    # one token more leaves the window: the loop swaps context like server.Do (server.go:160-172) - the ids up to there are the same ones, and
    # the whole run equals the checker's (which swaps by the Go lines; tests/test_context_swap.py)
    more = _sample_decode(product, shape, ctx, prompt, n + ctx // 2, 99, **kw)
    assert more[:n] == got
    assert more == _sample_decode(oracle, shape, ctx, prompt, n + ctx // 2, 99, **kw)


def test_sample_decode_int8(product, oracle):
    hp = make_hparams(**SHAPES["small"], ctx=48)
    outs = []
    for lib in (product, oracle):
        m = lib.NewSyntheticModel(hp, 1234).QuantizeQ8()
        c = m.NewContext(48, 16, False)
        outs.append(c.SampleDecode([1, 306, 1658, 278], 20, 40, 0.95, 0.8, 1.1, seed=3))
        c.free()
        m.free()
    assert outs[0] == outs[1]
