"""CPU (`-m "not gpu"`): the checker's SampleTopPTopK / SampleDecode against an independent numpy derivation and against the
properties the reference's algorithm implies (llama.go:455-707, server.go:127-217)."""
import numpy as np
import pytest

from llama_go_amd.mlapi import SHAPES, make_hparams, MLError
from sampler_ref import sample as np_sample, uniform


def _logits(rng, V, kind):
    x = rng.standard_normal(V).astype(np.float32) * 4
    if kind == "ties":        # coarse grid: many exactly equal values, also around the K-th place
        x = np.round(x * 2) / 2
    if kind == "neginf":
        x[rng.integers(0, V, V // 3)] = -np.inf
    if kind == "flat":
        x[:] = 1.25
    return x.astype(np.float32)


@pytest.mark.parametrize("kind", ["normal", "ties", "neginf", "flat"])
@pytest.mark.parametrize("V,topK,topP", [(512, 40, 0.95), (2048, 1, 0.95), (2048, 100, 0.5), (32000, 40, 0.95), (32000, 40, 1.0)])
def test_checker_sampler_matches_numpy(oracle, kind, V, topK, topP):
    rng = np.random.default_rng(V + topK)
    ctx = oracle.NewContext(1)
    for draw in range(4):
        lg = _logits(rng, V, kind)
        ring = [0] * 8 + [int(t) for t in rng.integers(0, V, 24)]
        tok, ids, probs = oracle.SampleTopPTopK(ctx, lg, ring, topK, topP, 0.8, 1.1, seed=77, draw=draw, debug=True)
        rtok, rids, rprobs = np_sample(lg, ring, topK, topP, 0.8, 1.1, 77, draw)
        assert ids == rids
        np.testing.assert_array_equal(probs, rprobs)  # both use libm exp in f64
        assert tok == rtok
        assert oracle.SampleTopPTopK(ctx, lg, ring, topK, topP, 0.8, 1.1, seed=77, draw=draw) == tok


def test_sampler_properties(oracle):
    rng = np.random.default_rng(5)
    ctx = oracle.NewContext(1)
    V = 1000
    lg = rng.standard_normal(V).astype(np.float32) * 3
    # topK = 1 is greedy on the penalised logits
    assert oracle.SampleTopPTopK(ctx, lg, [], 1, 0.95, 0.5, 1.1, seed=1) == int(np.argmax(lg))
    # the ring starts as zeros in the reference: id 0 is penalised until the zeros are overwritten (server.go:127-138)
    lg2 = lg.copy()
    lg2[0] = lg2.max() * 1.05 + 0.01
    assert oracle.SampleTopPTopK(ctx, lg2, [], 1, 0.95, 1.0, 1.5, seed=1) == 0
    assert oracle.SampleTopPTopK(ctx, lg2, [0, 0, 0], 1, 0.95, 1.0, 1.5, seed=1) != 0
    # negative logits are multiplied by the penalty (pushed further down), positive ones divided (llama.go:517-523)
    _, ids, probs = oracle.SampleTopPTopK(ctx, np.array([2.0, -1.0, 1.9, -1.1], np.float32), [0, 1], 4, 1.0, 1.0, 2.0, seed=3, debug=True)
    assert ids == [2, 0, 3, 1]  # 1.9, 2.0/2, -1.1, -1.0*2
    assert abs(float(probs.sum()) - 1.0) < 1e-6
    # same (seed, draw) -> same token; the draw index changes the uniforms
    picks = {oracle.SampleTopPTopK(ctx, lg, [], 40, 1.0, 2.0, 1.0, seed=9, draw=d) for d in range(32)}
    assert len(picks) > 3
    assert all(0.0 <= float(uniform(9, d, j)) < 1.0 for d in range(4) for j in range(50))
    # parameter errors are reported, not silently clamped (the reference would panic on logitsID[:topK], llama.go:567)
    for bad in (dict(topK=0), dict(topK=V + 1), dict(temp=0.0)):
        kw = dict(topK=40, topP=0.95, temp=0.8, repeatPenalty=1.1)
        kw.update(bad)
        with pytest.raises(MLError):
            oracle.SampleTopPTopK(ctx, lg, [], seed=1, **kw)


def test_sample_decode_loop(oracle):
    """SampleDecode == Eval + SampleTopPTopK driven from Python with the ring handling of server.go:127-217."""
    hp = make_hparams(**SHAPES["tiny"], ctx=32)
    m = oracle.NewSyntheticModel(hp, 1234)
    V = hp.vocabSize
    prompt = [1, 306 % V, 4658 % V, 278 % V]
    c1 = m.NewContext(32)
    got = c1.SampleDecode(prompt, 10, 40, 0.95, 0.8, 1.1, seed=42)
    c2 = m.NewContext(32)
    ring = [0] * 32
    pos = 0
    for t in prompt:
        ring[pos % 32] = t
        pos += 1
    lg = c2.Eval(prompt, 0)
    past = len(prompt)
    want = []
    mlctx = oracle.NewContext(1)
    for s in range(10):
        tok = oracle.SampleTopPTopK(mlctx, lg, ring, 40, 0.95, 0.8, 1.1, seed=42, draw=s)
        ring[pos % 32] = tok
        pos += 1
        want.append(tok)
        if s + 1 < 10:
            lg = c2.Eval([tok], past)
            past += 1
    assert got == want
    assert c1.SampleDecode(prompt, 10, 40, 0.95, 0.8, 1.1, seed=42) == got       # reproducible
    assert c1.SampleDecode(prompt, 10, 40, 0.95, 0.8, 1.1, seed=43) != got       # and seed-dependent
