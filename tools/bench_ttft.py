"""Time to first token vs prompt length (BASELINE.json configs[1] shape): one llama.Eval of N tokens at past = 0 on the 7B model,
host graph build and last-row logits D2H included.  usage: python tools/bench_ttft.py [--shape 7B] [--ns 2,4,8,9,16,32,64] [--int8]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from llama_go_amd.mlapi import SHAPES, load_product, make_hparams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="7B")
ap.add_argument("--ns", default="1,2,4,8,9,16,32,64")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--int8", action="store_true")
args = ap.parse_args()
ns = [int(x) for x in args.ns.split(",")]
prod = load_product()
hp = make_hparams(**SHAPES[args.shape], ctx=max(ns) + 8)
m = prod.NewSyntheticModel(hp, 1234)
if args.int8:
    m.QuantizeQ8()
c = m.NewContext(max(ns) + 8, 1)
rng = np.random.default_rng(0)
out = {}
for N in ns:
    toks = [int(t) for t in rng.integers(0, hp.vocabSize, N)]
    c.Eval(toks, 0)
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        lg = c.Eval(toks, 0)
        ts.append(time.perf_counter() - t0)
    out[N] = round(min(ts) * 1e3, 3)
print(json.dumps({"shape": args.shape + (" block-int8" if args.int8 else ""), "ms_per_eval_by_prompt_length": out,
                  "env": {k: v for k, v in os.environ.items() if k.startswith("LLAMAHIP_")}}))
