"""Writes tests/golden/7b_seed1234_ids.json: the greedy ids of the headline workload (LLaMA-7B fp32, synthetic weights seed 1234, the
fixed 8-token prompt, context 128) from the single-GPU device loop, TOGETHER with the checker's verdict on them (the CPU restatement
decodes the same prompt; ids must be equal on every step).  bench.py --gpus N compares stream 0 of the layer-sharded pipeline with
this file.  --int8: the same workload on block-int8 weights (BASELINE config 4; the checker decodes on the dequantised
weights) -> tests/golden/7b_seed1234_int8_ids.json.  tests/test_gpu_llama.py::test_headline_workload_at_full_depth reads both files.
Needs an MI355X.  usage: python tools/make_golden_ids.py [--n 100] [--oracle-steps 100] [--int8]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
import numpy as np  # noqa: E402
from llama_go_amd.mlapi import PROMPT, SHAPES, MLLib, decode_greedy_resident, load_product, make_hparams, usable_threads  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--oracle-steps", type=int, default=100)
ap.add_argument("--int8", action="store_true")
args = ap.parse_args()
FILE = "7b_seed1234_int8_ids.json" if args.int8 else "7b_seed1234_ids.json"
prod = load_product()
hp = make_hparams(**SHAPES["7B"], ctx=128)
m = prod.NewSyntheticModel(hp, 1234)
if args.int8:
    m.QuantizeQ8()
c = m.NewContext(128, 1)
first = int(np.argmax(c.Eval(PROMPT, 0)))
toks, _ = decode_greedy_resident(c, first, len(PROMPT), args.n - 1)
ids = [first] + toks
c.free()
m.free()
orc = MLLib(os.path.join(ROOT, "oracle", "liboracle.so"))
om = orc.NewSyntheticModel(hp, 1234)
if args.int8:
    om.QuantizeQ8()
oc = om.NewContext(128, usable_threads(), False)
no = min(args.oracle_steps, args.n)
otoks, ologits = oc.GreedyDecode(PROMPT, no)
oc.free()
om.free()
srt = np.sort(ologits, axis=-1)
out = {"workload": "LLaMA-7B " + ("block-int8 (format of csrc/kernels_q8.h; the checker decodes on the dequantised weights)" if args.int8 else "fp32") + ", synthetic weights seed 1234, prompt [1, 306, 4658, 278, 6593, 310, 2834, 338], context 128, greedy",
       "ids": ids, "oracle_ids": [int(t) for t in otoks], "oracle_ids_match": ids[:no] == [int(t) for t in otoks], "oracle_steps": no,
       "min_top2_margin_rel": float(((srt[:, -1] - srt[:, -2]) / np.abs(ologits).max(axis=-1)).min()),
       "generator": "tools/make_golden_ids.py (ids[0] = argmax of the prompt's last row; ids[i] = id produced at position 7 + i)"}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", FILE), "w"), indent=1)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", FILE), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("oracle_ids_match", "oracle_steps", "min_top2_margin_rel")}), ids[:20])
