"""What one tick of the C pipeline scheduler costs on ONE GPU (world of one rank): the same 4 greedy streams through lh_pipeline_run
(a) without a communicator (the produced token id is copied device-to-device), (b) with RCCL: the id travels last stage -> first
stage as a grouped ncclSend + ncclRecv to self on the context's stream, and (c) the single-context resident decode loop (hipGraph
replay) for reference.  (b) - (a) = the enqueue + execution cost of one RCCL p2p group per tick as this library issues it; the
xGMI hop itself cannot be measured on a one-GPU box.  usage: python tools/bench_pipeline_overhead.py [--shape 7B] [--steps 32]"""
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
ap.add_argument("--steps", type=int, default=32)
ap.add_argument("--pods", type=int, default=4)
args = ap.parse_args()
prod = load_product()
hp = make_hparams(**SHAPES[args.shape], ctx=128)
m = prod.NewSyntheticModel(hp, 1234)
prompt = [t % hp.vocabSize for t in PROMPT]
out = {"shape": args.shape, "pods": args.pods, "steps": args.steps}
for name, cid in (("no_comm", None), ("rccl_self", comm_unique_id(prod))):
    pl = Pipeline(m, 128, args.pods, 0, 1, comm_id=cid)
    pl.run([prompt] * args.pods, 2)
    t0 = time.perf_counter()
    pl.run(None, args.steps)
    dt = time.perf_counter() - t0
    out[name] = {"us_per_tick": round(dt / (args.steps * args.pods) * 1e6, 2), "tokens_per_s": round(args.steps * args.pods / dt, 2), "tokens_stream0": pl.tokens(0)[:6]}
    pl.free()
c = m.NewContext(128, 1)
first = int(np.argmax(c.Eval(prompt, 0)))
decode_greedy_resident(c, first, len(prompt), 2)
t0 = time.perf_counter()
decode_greedy_resident(c, first, len(prompt), args.steps)
dt = time.perf_counter() - t0
out["resident_graph_loop"] = {"us_per_step": round(dt / args.steps * 1e6, 2), "tokens_per_s": round(args.steps / dt, 2)}
out["rccl_group_us_per_tick"] = round(out["rccl_self"]["us_per_tick"] - out["no_comm"]["us_per_tick"], 2)
c.free()
m.free()
print(json.dumps(out))
