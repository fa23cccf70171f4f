"""A/B driver of the decode attention folded into the wq|wk|wv launch (tools/fused_attention_tail.h; not in the product since the A/B, profiles/r04_fused_attention_ab.txt; LLAMAHIP_FUSED_ATTN is read once per process by a build that has it): the resident greedy loop of
the 7B model for --steps tokens x --runs runs (context 128: the loop swaps when the window is full), the hash of every run's ids, the hash of the
final logits, and tokens/s.  Run it once per setting of the variable and compare the lines (tools/gpu/r4_fused.sh does).
usage: [LLAMAHIP_FUSED_ATTN=1] python tools/check_fused_attn.py [--int8] [--steps 1000] [--runs 3] [--ctx 128]"""
import argparse, hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from llama_go_amd.mlapi import SHAPES, load_product, make_hparams, decode_greedy_resident

ap = argparse.ArgumentParser()
ap.add_argument("--int8", action="store_true"); ap.add_argument("--steps", type=int, default=1000); ap.add_argument("--runs", type=int, default=3)
ap.add_argument("--ctx", type=int, default=128); ap.add_argument("--shape", default="7B"); ap.add_argument("--layers", type=int, default=0)
args = ap.parse_args()
prod = load_product()
kw = dict(SHAPES[args.shape])
if args.layers:
    kw["layers"] = args.layers
hp = make_hparams(**kw, ctx=args.ctx)
m = prod.NewSyntheticModel(hp, 1234)
if args.int8:
    m.QuantizeQ8()
prompt = [1, 15043, 3186, 29892, 445, 338, 263, 1243]
runs = []
for r in range(args.runs):
    c = m.NewContext(args.ctx, 1)
    lg = c.Eval(prompt, 0)
    first = int(np.argmax(lg))
    decode_greedy_resident(c, first, len(prompt), 8)          # warm the graphs (re-run below from the same state: positions are overwritten)
    t0 = time.perf_counter()
    ids, lg2 = decode_greedy_resident(c, first, len(prompt), args.steps, want_logits=True)
    dt = time.perf_counter() - t0
    runs.append({"ids_sha": hashlib.sha256(np.asarray(ids, dtype=np.uint32).tobytes()).hexdigest()[:16],
                 "logits_sha": hashlib.sha256(lg2.tobytes()).hexdigest()[:16], "tok_s": round(args.steps / dt, 2), "first_ids": ids[:6]})
    c.free()
print(json.dumps({"fused": os.environ.get("LLAMAHIP_FUSED_ATTN", ""), "int8": args.int8, "steps": args.steps, "ctx": args.ctx, "runs": runs}))
