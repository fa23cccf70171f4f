"""Prefill measurement (BASELINE.json configs[2]): LLaMA-13B shape fp32, one Eval of N tokens at past = 0, context N.
Reports TFLOP/s against the fp32 MFMA peak (157.3 TF).  usage: python tools/bench_prefill.py [--shape 13B] [--n 1024] [--layers L]"""
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
ap.add_argument("--shape", default="13B")
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--layers", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--int8", action="store_true", help="block-int8 weight matrices (config 4): the dequantising GEMM")
args = ap.parse_args()
prod = load_product()
kw = dict(SHAPES[args.shape])
if args.layers:
    kw["layers"] = args.layers
hp = make_hparams(**kw, ctx=args.n)
m = prod.NewSyntheticModel(hp, 1234)
if args.int8:
    m.QuantizeQ8()
c = m.NewContext(args.n, 1)
d, L, V, F, N = hp.embdSize, hp.layersCount, hp.vocabSize, m.ffSize, args.n
rng = np.random.default_rng(0)
toks = [int(t) for t in rng.integers(0, V, N)]
c.Eval(toks, 0)  # warm-up (allocations, kernel attributes)
ts = []
for _ in range(args.reps):
    t0 = time.perf_counter()
    lg = c.Eval(toks, 0)
    ts.append(time.perf_counter() - t0)
dt = min(ts)
# executed flops: the layer matmuls for all N rows, the lm_head for the ONE row llama.Eval reads (llama.go:394-401; the reference
# itself multiplies all N rows, llama.go:384 — LH_GRAPH_LAST_ROW_LOGITS), attention over the causal half only
flops_layers = 2.0 * N * L * (4 * d * d + 3 * d * F)
flops_w = flops_layers + 2.0 * V * d
flops_ref = flops_layers + 2.0 * N * V * d            # what the reference's graph multiplies
flops_a = L * 4.0 * N * N * d                          # full (unmasked) score block, as the reference computes it
print(json.dumps({"shape": args.shape + (" block-int8" if args.int8 else ""), "layers": L, "N": N, "seconds": round(dt, 4), "tflop_weights_executed": round(flops_w / 1e12, 2),
                  "tflop_weights_reference_graph": round(flops_ref / 1e12, 2), "tflop_attention_full": round(flops_a / 1e12, 2),
                  "TFLOPs_per_s_weights_executed": round(flops_w / dt / 1e12, 1),
                  "frac_of_157.3TF_fp32_mfma_peak": round(flops_w / dt / 157.3e12, 3),
                  "TFLOPs_per_s_reference_graph_equivalent": round((flops_ref + flops_a) / dt / 1e12, 1),
                  "logit_checksum": float(np.abs(lg).sum())}))
