"""What one tick of the C pipeline scheduler costs the HOST on ONE GPU (world of one rank), against the GPU time of the same work.
A SHORT stage (--layers 4: about what an 8-rank 7B shard holds) makes host cost visible: a 32-layer stage hides anything under 4.3 ms.
  resident_graph_loop  lh_llama_decode_greedy: multi-step hipGraph replay, no host work between steps = the GPU time of one step
  no_comm / rccl_self  lh_pipeline_run, `--pods` streams with `--max-rows` streams per tick (1 = every stream its own tick): every tick
                       = one hipGraphLaunch of the group's captured tick (+ with RCCL: one grouped ncclSend + ncclRecv to self of the
                       ids on the context's stream; the xGMI hop itself cannot be measured on a one-GPU box)
  tick_minus_gpu_us    per tick, for max-rows 1: what the scheduler + graph launch (+ RCCL group) add over the bare GPU time
usage: python tools/bench_pipeline_overhead.py [--shape 7B] [--layers 4] [--steps 64] [--pods 4] [--max-rows 1]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from llama_go_amd.mlapi import PROMPT, SHAPES, Pipeline, comm_unique_id, decode_greedy_resident, load_product, make_hparams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="7B")
ap.add_argument("--layers", type=int, default=4)
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--pods", type=int, default=4)
ap.add_argument("--max-rows", type=int, default=1)
args = ap.parse_args()
prod = load_product()
kw = dict(SHAPES[args.shape])
if args.layers:
    kw["layers"] = args.layers
ctx_size = max(128, len(PROMPT) + args.steps + 8)
hp = make_hparams(**kw, ctx=ctx_size)
m = prod.NewSyntheticModel(hp, 1234)
prompt = [t % hp.vocabSize for t in PROMPT]
out = {"shape": args.shape, "layers": kw["layers"], "pods": args.pods, "max_rows": args.max_rows, "steps": args.steps}
for name, cid in (("no_comm", None), ("rccl_self", comm_unique_id(prod))):
    pl = Pipeline(m, ctx_size, args.pods, 0, 1, comm_id=cid, max_rows=args.max_rows)
    pl.run([prompt] * args.pods, 3)
    best = None
    for _ in range(3):
        pl.run([prompt] * args.pods, 3)
        t0 = time.perf_counter()
        pl.run(None, args.steps)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    ticks = args.steps * pl.groups
    out[name] = {"us_per_tick": round(best / ticks * 1e6, 2), "ticks": ticks, "rows_per_tick": args.pods // pl.groups,
                 "tokens_per_s": round(args.steps * args.pods / best, 2), "tokens_stream0": pl.tokens(0)[:6]}
    pl.free()
c = m.NewContext(ctx_size, 1)
first = int(np.argmax(c.Eval(prompt, 0)))
decode_greedy_resident(c, first, len(prompt), 8)
best = None
for _ in range(3):
    t0 = time.perf_counter()
    decode_greedy_resident(c, first, len(prompt), args.steps)
    dt = time.perf_counter() - t0
    best = dt if best is None else min(best, dt)
out["resident_graph_loop"] = {"us_per_step": round(best / args.steps * 1e6, 2), "tokens_per_s": round(args.steps / best, 2)}
out["rccl_group_us_per_tick"] = round(out["rccl_self"]["us_per_tick"] - out["no_comm"]["us_per_tick"], 2)
if out["no_comm"]["rows_per_tick"] == 1:
    out["tick_minus_gpu_us"] = {k: round(out[k]["us_per_tick"] - out["resident_graph_loop"]["us_per_step"], 2) for k in ("no_comm", "rccl_self")}
c.free()
m.free()
print(json.dumps(out))
