"""The reference's --pods on ONE GPU (pkg/server/server.go:88-101): P independent greedy streams whose decode steps share one pass over the
weights (lh_batch through the pipeline scheduler, world = 1).  Aggregate tokens/s and ms per tick per P, ids checked against a single stream.
usage: python tools/bench_pods.py [--shape 7B] [--pods 1,2,4,8,16,32,48,64] [--steps 32] [--int8] [--max-rows 0]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llama_go_amd.mlapi import PROMPT, SHAPES, Pipeline, load_product, make_hparams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="7B")
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--pods", default="1,2,4,8,16,32,48,64")
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--int8", action="store_true")
ap.add_argument("--max-rows", type=int, default=0)
args = ap.parse_args()
prod = load_product()
kw = dict(SHAPES[args.shape])
if args.layers:
    kw["layers"] = args.layers
ctx_size = max(128, len(PROMPT) + args.steps + 8)
hp = make_hparams(**kw, ctx=ctx_size)
m = prod.NewSyntheticModel(hp, 1234)
if args.int8:
    m.QuantizeQ8()
prompt = [t % hp.vocabSize for t in PROMPT]
out, ref = {}, None
for P in [int(x) for x in args.pods.split(",")]:
    pl = Pipeline(m, ctx_size, P, 0, 1, max_rows=args.max_rows)
    pl.run([prompt] * P, 3)
    t0 = time.perf_counter()
    pl.run(None, args.steps)
    dt = time.perf_counter() - t0
    ids = [pl.tokens(i) for i in range(P)]
    groups = pl.groups
    pl.free()
    ref = ref or ids[0]
    out[P] = {"tokens_per_s": round(P * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 4), "groups": groups, "rows_per_tick": P // groups,
              "ids_equal_single_stream": all(t == ref for t in ids)}
m.free()
print(json.dumps({"shape": args.shape + (" block-int8" if args.int8 else ""), "layers": kw["layers"], "steps": args.steps, "by_pods": out}))
