"""Quick same-box A/B of the resident greedy loop (7B fp32 / --int8): tokens/s of K steps behind the 8-token prompt, best of R runs, and the ids' checksum.
usage: [ENV=...] python tools/decode_quick.py [--steps 64] [--reps 5] [--int8]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from llama_go_amd.mlapi import PROMPT, SHAPES, decode_greedy_resident, load_product, make_hparams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--int8", action="store_true")
ap.add_argument("--shape", default="7B")
args = ap.parse_args()
prod = load_product()
hp = make_hparams(**SHAPES[args.shape], ctx=128)
m = prod.NewSyntheticModel(hp, 1234)
if args.int8:
    m.QuantizeQ8()
c = m.NewContext(128, 1)
best, ids = 1e9, None
for r in range(args.reps + 1):
    lg = c.Eval(PROMPT, 0)
    first = int(lg.argmax())
    c.Sync() if hasattr(c, "Sync") else None
    t0 = time.perf_counter()
    toks, _ = decode_greedy_resident(c, first, len(PROMPT), args.steps, want_logits=False)
    dt = time.perf_counter() - t0
    if r:
        best = min(best, dt)
    ids = toks
print(f"{args.shape}{' int8' if args.int8 else ''} PREFETCH={os.environ.get('LLAMAHIP_PREFETCH', '0')}: {args.steps / best:.2f} tok/s  {best / args.steps * 1e3:.4f} ms/token  ids checksum {sum((i + 1) * t for i, t in enumerate(ids)) % 1000003}")
