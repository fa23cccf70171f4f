"""CPU-only: the checker's own top-2 logit margins, at EVERY greedy step, for the configurations whose GPU tests assert token ids.
A GPU test may only assert ids where the checker itself is not at a near-tie (margin > 10 x the 1e-4 tolerance); this script is how the
seeds in tests/test_gpu_batch.py and tests/test_gpu_llama.py were chosen.  usage: python tools/check_test_margins.py [name-substring]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from llama_go_amd.mlapi import SHAPES, MLLib, make_hparams  # noqa: E402

orc = MLLib(os.path.join(ROOT, "oracle", "liboracle.so"))
flt = sys.argv[1] if len(sys.argv) > 1 else ""


def margin(lg):
    s = np.sort(lg, axis=-1)
    return float(((s[..., -1] - s[..., -2]) / np.abs(lg).max(axis=-1)).min())


def streams(hp, seed, prompts, n_predict, ctx, int8=False):
    om = orc.NewSyntheticModel(hp, seed)
    if int8:
        om.QuantizeQ8()
    worst = 1e9
    for pr in prompts:
        oc = om.NewContext(ctx, 16, False)
        _, lg = oc.GreedyDecode(pr, n_predict)
        oc.free()
        worst = min(worst, margin(lg))
    om.free()
    return worst


def report(name, m):
    print(f"{'OK  ' if m > 2.5e-4 else 'TIE '} {m:.2e}  {name}", flush=True)


import test_gpu_batch as tb  # noqa: E402
import test_gpu_llama as tl  # noqa: E402


def params(fn):
    for mk in getattr(fn, "pytestmark", []):
        if mk.name == "parametrize":
            yield mk.args[0], mk.args[1]


if ("batched_decode_equals" in flt or not flt) and "--search" not in sys.argv:
    for _, cases in params(tb.test_batched_decode_equals_every_stream_alone):
        for ci, (shape, layers, int8, lengths, seed) in enumerate(cases):
            if "--from" in sys.argv and ci < int(sys.argv[sys.argv.index("--from") + 1]):
                continue
            kw = dict(SHAPES[shape])
            if layers:
                kw["layers"] = layers
            hp = make_hparams(**kw, ctx=32)
            rng = np.random.default_rng(len(lengths) * 131 + int(int8))
            prompts = tb.make_prompts(rng, kw["vocab"], lengths)
            report(f"batched_decode_equals [{ci}] {shape} layers={layers} int8={int8} rows={len(lengths)} seed={seed}", streams(hp, seed, prompts, 5, 32, int8))
if ("odd_shapes" in flt or not flt) and "--search" not in sys.argv:
    for _, cases in params(tl.test_odd_shapes_match_oracle):
        for kw, ctx, n_prompt in cases:
            hp = make_hparams(**kw, ctx=ctx)
            rng = np.random.default_rng(n_prompt + kw["embd"])
            report(f"odd_shapes_match_oracle embd={kw['embd']} n={n_prompt}", streams(hp, 99, [[int(t) for t in rng.integers(0, kw["vocab"], n_prompt)]], 6, ctx))
    for _, cases in params(tl.test_odd_shapes_block_int8):
        for kw, ctx, n_prompt in cases:
            hp = make_hparams(**kw, ctx=ctx)
            rng = np.random.default_rng(n_prompt + kw["embd"] + 1)
            report(f"odd_shapes_block_int8 embd={kw['embd']} n={n_prompt}", streams(hp, 99, [[int(t) for t in rng.integers(0, kw["vocab"], n_prompt)]], 5, ctx, True))
    for _, cases in params(tb.test_batched_decode_odd_shapes):
        for (kw,) in [(c,) for c in cases]:
            hp = make_hparams(**kw, ctx=40)
            rng = np.random.default_rng(kw["embd"])
            report(f"batched_decode_odd_shapes embd={kw['embd']}", streams(hp, 3, tb.make_prompts(rng, kw["vocab"], [5, 1, 8]), 5, 40))
if ("7b_slice" in flt or not flt) and "--search" not in sys.argv:
    for _, cases in params(tl.test_7b_shape_slice_short_prompts_match_oracle):
        for n_prompt in cases:
            kw = dict(SHAPES["7B"]); kw["layers"] = 2
            ctx = 64 if n_prompt <= 56 else (128 if n_prompt <= 120 else 192)
            hp = make_hparams(**kw, ctx=ctx)
            rng = np.random.default_rng(100 + n_prompt)
            report(f"7b_shape_slice_short_prompts n={n_prompt}", streams(hp, 1234, [[int(t) for t in rng.integers(0, kw["vocab"], n_prompt)]], 3, ctx))
if ("long_context" in flt or not flt) and "--search" not in sys.argv:
    kw = dict(vocab=515, embd=640, mult=32, heads=5, layers=2)
    hp = make_hparams(**kw, ctx=320)
    rng = np.random.default_rng(11)
    report("batched_decode_long_context", streams(hp, 8, tb.make_prompts(rng, kw["vocab"], [260, 3, 127, 130]), 4, 320))
if ("pipeline_groups" in flt or not flt) and "--search" not in sys.argv:
    for int8 in (False, True):
        hp = make_hparams(**SHAPES["small"], ctx=40)
        rng = np.random.default_rng(21)
        report(f"pipeline_groups int8={int8}", streams(hp, 17, tb.make_prompts(rng, hp.vocabSize, [5, 1, 8, 2, 12, 3]), 6, 40, int8))

# --search: model seeds for the batched-decode configurations whose default seed leaves a near-tie (usage: ... --search <config index> <n seeds>)
if "--search" in sys.argv:
    i = sys.argv.index("--search")
    ci, ns = int(sys.argv[i + 1]), int(sys.argv[i + 2])
    cases = [c for _, cs in params(tb.test_batched_decode_equals_every_stream_alone) for c in cs]
    shape, layers, int8, lengths = cases[ci][:4]
    kw = dict(SHAPES[shape])
    if layers:
        kw["layers"] = layers
    hp = make_hparams(**kw, ctx=32)
    rng = np.random.default_rng(len(lengths) * 131 + int(int8))
    prompts = tb.make_prompts(rng, kw["vocab"], lengths)
    for seed in range(5000, 5000 + ns):
        m = streams(hp, seed, prompts, 5, 32, int8)
        print(f"config {ci} seed {seed}: {m:.2e}", flush=True)
        if m > 5e-4:
            break
