"""One Eval of an n-token chunk behind a cache of `past` tokens (a long prompt fed in pieces, or the next turn of a long conversation):
LLaMA-7B-shaped layers, fp32.  The prefill attention has few, long query blocks there (64 queries behind 1984 keys: 32 blocks of 32
steps for 512 workgroup slots) - the case its key-range parts exist for (kernels_attn.h "Balance").
Wall time of the Eval (host graph build and last-row logits D2H included), median of --reps; kernel times: run it under rocprofv3.
usage: python tools/bench_chunked_prefill.py [--layers 8] [--cases 64:1984 128:1920 256:1792 512:512 512:0 1024:0]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from llama_go_amd.mlapi import SHAPES, load_product, make_hparams
ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="7B")
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--cases", nargs="+", default=["64:1984", "128:1920", "256:1792", "512:512", "512:0", "1024:0"])
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
prod = load_product()
cases = [tuple(int(v) for v in c.split(":")) for c in args.cases]
ctx_size = max(n + p for n, p in cases)
kw = dict(SHAPES[args.shape]); kw["layers"] = args.layers
hp = make_hparams(**kw, ctx=ctx_size)
m = prod.NewSyntheticModel(hp, 1234)
rng = np.random.default_rng(0)
toks = [int(t) for t in rng.integers(0, hp.vocabSize, ctx_size)]
out = []
for n, past in cases:
    c = m.NewContext(ctx_size, 1)
    for s in range(0, past, 512):                      # the cache behind the chunk
        c.Eval(toks[s:min(s + 512, past)], s)
    c.Eval(toks[past:past + n], past)                  # warm-up: scratch, kernel attributes
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter(); c.Eval(toks[past:past + n], past); ts.append(time.perf_counter() - t0)
    out.append({"n": n, "past": past, "ms": round(sorted(ts)[len(ts) // 2] * 1e3, 3)})
    c.free()
print(json.dumps({"shape": args.shape, "layers": args.layers, "evals": out}))
