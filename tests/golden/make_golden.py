"""Generates the committed known-answer vectors (tests/golden/*.npz) from the oracle.  Run from the repo root:
    python tests/golden/make_golden.py
The reference has no golden vectors of its own and cannot be executed here (Go, no toolchain), so these
KATs pin OUR restatement (already cross-checked by tests/test_oracle.py against an independent float64 derivation and the
reference's own vdot C source) against silent drift in later sessions."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from llama_go_amd.mlapi import MLLib, SHAPES, make_hparams  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
orc = MLLib(os.path.join(ROOT, "oracle", "liboracle.so"))

# ---- end-to-end: tiny model, prompt of 8, 8 greedy steps (scalar pure-Go order, 1 thread)
ctx_size, seed, prompt = 32, 1234, [1, 5, 9, 200, 17, 3, 44, 100]
hp = make_hparams(**SHAPES["tiny"], ctx=ctx_size)
m = orc.NewSyntheticModel(hp, seed)
c = m.NewContext(ctx_size, 1, False)
toks, lg = c.GreedyDecode(prompt, 8)
w2 = orc.read(None, m.tensor("layers.1.feed_forward.w2.weight")).reshape(-1)[:64].copy()
np.savez_compressed(os.path.join(HERE, "tiny_eval.npz"), ctx=ctx_size, seed=seed, prompt=np.array(prompt), tokens=np.array(toks), logits=lg, w2_head=w2)
c.free()
m.free()

# ---- per-op KATs
rng = np.random.default_rng(2024)
ctx = orc.NewContext(1)


def leaf(arr):
    arr = np.asarray(arr, np.float32)
    return orc.NewTensor(ctx, tuple(reversed(arr.shape)), data=arr)


def run(t):
    g = orc.NewGraph()
    orc.BuildForwardExpand(g, t)
    orc.GraphCompute(ctx, g)
    out = orc.read(ctx, t).copy()
    orc.FreeGraph(g)
    return out


d = {}
d["mm_w"] = (rng.standard_normal((6, 4096)) / 64).astype(np.float32)
d["mm_x"] = rng.standard_normal((2, 4096)).astype(np.float32)
d["mm_y"] = run(orc.MulMat(ctx, leaf(d["mm_w"]), leaf(d["mm_x"])))
d["rn_x"] = (rng.standard_normal((2, 512)) * 2).astype(np.float32)
d["rn_g"] = (1 + 0.1 * rng.standard_normal(512)).astype(np.float32)
cur = orc.RMSNorm(ctx, leaf(d["rn_x"]))
d["rn_y"] = run(orc.Mul(ctx, orc.Repeat(ctx, leaf(d["rn_g"]), cur), cur))
d["rope_x"] = rng.standard_normal((3, 2, 128)).astype(np.float32)
d["rope_y0"] = run(orc.Rope(ctx, leaf(d["rope_x"]), 3, 128, 0))
d["rope_y1"] = run(orc.Rope(ctx, leaf(d["rope_x"]), 1, 128, 1))
d["sm_x"] = rng.standard_normal((2, 3, 5)).astype(np.float32)  # [H=2][N=3][T=5], past = 2 (T < 8)
d["sm_scale"] = np.float32(1.0 / np.sqrt(128.0))
d["sm_y"] = run(orc.SoftMax(ctx, orc.DiagMaskInf(ctx, orc.Scale(ctx, leaf(d["sm_x"]), orc.NewFP32(ctx, float(d["sm_scale"]))), 2)))
d["silu_x"] = (rng.standard_normal((1, 300)) * 5).astype(np.float32)
d["silu_y"] = run(orc.Silu(ctx, leaf(d["silu_x"])))
np.savez_compressed(os.path.join(HERE, "ops.npz"), **d)
print("wrote", os.listdir(HERE))
