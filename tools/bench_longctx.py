"""Decode step time vs position (long-context check): LLaMA-7B fp32, prefill P tokens, then 16 resident decode steps.
Then the same for 8 pods in one weight pass (lh_batch ticks: per-row split-T attention), every pod behind its own P-token prompt.
usage: python tools/bench_longctx.py [--past 1000] [--pods 8]
(Under rocprofv3 give ONE position per invocation: with four or more contexts captured one after the other in one traced process - each with its 8-step decode graphs of
~1800 kernel nodes - the profiler's tool library segfaults inside the fourth context's first graph launch; the same sequence runs clean without the profiler, also under
MALLOC_CHECK_=3, and every single position runs clean with it: profiles/r06_longctx.txt.)"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from llama_go_amd.mlapi import SHAPES, load_product, make_hparams, decode_greedy_resident, profile_decode
ap = argparse.ArgumentParser(); ap.add_argument("--past", type=int, nargs="+", default=[8, 120, 500, 1000, 2000]); ap.add_argument("--shape", default="7B")
ap.add_argument("--pods", type=int, default=8)
args = ap.parse_args()
prod = load_product()
ctx_size = max(args.past) + 32
hp = make_hparams(**SHAPES[args.shape], ctx=ctx_size)
m = prod.NewSyntheticModel(hp, 1234)
rng = np.random.default_rng(0)
out = []
for P in args.past:
    c = m.NewContext(ctx_size, 1)
    toks = [int(t) for t in rng.integers(0, hp.vocabSize, P)]
    lg = c.Eval(toks, 0)
    first = int(np.argmax(lg))
    decode_greedy_resident(c, first, P, 2)
    t0 = time.perf_counter(); decode_greedy_resident(c, first, P, 16); dt = (time.perf_counter() - t0) / 16
    prof = {k["name"]: round(k["avg_us"], 2) for k in profile_decode(c, first, P, 2)}
    out.append({"past": P, "ms_per_token": round(dt * 1e3, 4), "tok_s": round(1 / dt, 1), "attention_us": prof.get("attention"), "attention_split_us": prof.get("attention_split"), "attention_combine_us": prof.get("attention_combine")})
    c.free()
print(json.dumps(out))
if args.pods > 1:
    from llama_go_amd.mlapi import Pipeline
    outp = []
    for P in [p for p in args.past if p <= 1000]:
        prompts = [[int(t) for t in rng.integers(0, hp.vocabSize, P)] for _ in range(args.pods)]
        pl = Pipeline(m, ctx_size, args.pods, 0, 1)
        pl.run(prompts, 3)
        t0 = time.perf_counter(); pl.run(None, 16); dt = (time.perf_counter() - t0) / 16
        pl.free()
        outp.append({"past": P, "pods": args.pods, "ms_per_tick": round(dt * 1e3, 4), "tok_s": round(args.pods / dt, 1)})
    print(json.dumps(outp))
