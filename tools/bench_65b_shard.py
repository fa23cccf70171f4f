"""BASELINE.json configs[4] on the hardware we can reach: ONE rank's share of LLaMA-65B fp32 layer-sharded over 8 GPUs
(10 of 80 layers = 32.4 GB of weights; the last rank additionally holds the 1.05 GB lm_head).  Times the stage
(lh_llama_stage) in steady-state decode on a single MI355X and PROJECTS the 8-GPU numbers from it — projections, not
measurements: gpurun exposes one GPU.  usage: python tools/bench_65b_shard.py [--stage first|middle|last]"""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from llama_go_amd.mlapi import SHAPES, load_product, make_hparams

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=32); ap.add_argument("--ranks", type=int, default=8); ap.add_argument("--shape", default="65B")
args = ap.parse_args()
prod = load_product()
hp = make_hparams(**SHAPES[args.shape], ctx=128)
L, d, V, R = hp.layersCount, hp.embdSize, hp.vocabSize, args.ranks
res = {}
tstream = torch.cuda.Stream(); torch.cuda.set_stream(tstream)
prod.lib.llamago_SetStream(C.c_void_p(tstream.cuda_stream))
for name, rank in (("first", 0), ("middle", R // 2), ("last", R - 1)):
    l0, l1 = rank * L // R, (rank + 1) * L // R
    m = prod.NewSyntheticModel(hp, 1234, l0, l1)
    c = m.NewContext(128, 1)
    xin = torch.randn(d, device="cuda") * 0.5
    xout = torch.empty(d, device="cuda")
    tok = torch.zeros(1, dtype=torch.int32, device="cuda")
    tokh = (C.c_uint32 * 1)(17)
    def stage(past):
        rc = prod.lib.llamago_Stage(c.h, tokh if rank == 0 else None, None, None if rank == 0 else C.c_void_p(xin.data_ptr()),
                                    None if rank == R - 1 else C.c_void_p(xout.data_ptr()), 1, past, None, C.c_void_p(tok.data_ptr()) if rank == R - 1 else None)
        assert rc == 0, prod.last_error()
    for s in range(4): stage(8 + s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(args.steps): stage(8 + (s % 100))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / args.steps
    F = m.ffSize
    wbytes = 4 * ((l1 - l0) * (4 * d * d + 3 * d * F + 2 * d) + (V * d + d if rank == R - 1 else 0) + (d if rank == 0 else 0))
    res[name] = {"layers": [l0, l1], "ms_per_token": round(dt * 1e3, 4), "weights_GB": round(wbytes / 1e9, 2), "stage_TBps": round(wbytes / dt / 1e12, 3)}
    c.free(); m.free()
stage_ms = [res["first"]["ms_per_token"]] + [res["middle"]["ms_per_token"]] * (R - 2) + [res["last"]["ms_per_token"]]
hop_ms = 0.02
out = {"config": "LLaMA-65B fp32, 80 layers sharded 10 per rank over 8 ranks (one rank measured at a time on ONE MI355X)", "stages": res,
       "PROJECTED_single_stream_tok_s_8gpu": round(1e3 / (sum(stage_ms) + R * hop_ms), 2),
       "PROJECTED_aggregate_tok_s_8gpu_8pods": round(1e3 / (max(stage_ms) + hop_ms), 2),
       "note": "projections from per-stage times + an assumed 20 us RCCL p2p hop; the 8-GPU run itself is the driver's"}
print(json.dumps(out))
