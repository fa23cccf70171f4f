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
# ---- the same stage through the product's scheduler: lh_pipeline_run with a REAL RCCL communicator (world of one: the ids travel last
# stage -> first stage as a grouped ncclSend / ncclRecv to self per tick), one stream per tick and four streams per tick (one weight pass
# for the four).  A 10-layer 65B-shape whole model = a stage + the embedding + the lm_head (what the first and the last rank hold together).
from llama_go_amd.mlapi import PROMPT, Pipeline, comm_unique_id
kw10 = dict(SHAPES[args.shape]); kw10["layers"] = (L + R - 1) // R
hp10 = make_hparams(**kw10, ctx=128)
m10 = prod.NewSyntheticModel(hp10, 1234)
prompt = [t % V for t in PROMPT]
ticks = {}
for pods in (1, 4):
    pl = Pipeline(m10, 128, pods, 0, 1, comm_id=comm_unique_id(prod))
    pl.run([prompt] * pods, 3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pl.run(None, args.steps)
    dtt = (time.perf_counter() - t0) / args.steps
    ticks[pods] = {"ms_per_tick": round(dtt * 1e3, 4), "rows_per_tick": pods, "groups": pl.groups, "tokens_stream0": pl.tokens(0)[:4]}
    pl.free()
m10.free()
stage_ms = [res["first"]["ms_per_token"]] + [res["middle"]["ms_per_token"]] * (R - 2) + [res["last"]["ms_per_token"]]
# per hop: what a tick of the scheduler adds over the bare GPU time of its stage WITH the RCCL group, measured on a 4-layer 7B stage on this
# hardware (profiles/r03_pipeline_tick_overhead_4layers.jsonl: +26 us) - the xGMI flight time of a 32 KB message is not in it (one GPU per box)
hop_ms = 0.026
out = {"config": "LLaMA-65B fp32, 80 layers sharded 10 per rank over 8 ranks (one rank measured at a time on ONE MI355X)", "stages": res,
       "stage_through_lh_pipeline_run_rccl_world_of_one": ticks,
       "PROJECTED_single_stream_tok_s_8gpu": round(1e3 / (sum(stage_ms) + R * hop_ms), 2),
       "PROJECTED_aggregate_tok_s_8gpu_8pods_one_row_per_tick": round(1e3 / (max(stage_ms) + hop_ms), 2),
       "PROJECTED_aggregate_tok_s_8gpu_32pods_four_rows_per_tick": round(4e3 / (ticks[4]["ms_per_tick"] + hop_ms), 2),
       "note": "PROJECTIONS from per-stage times measured on one GPU + the measured per-tick scheduler / RCCL-group cost; the xGMI hop itself and the 8-GPU "
               "run are the driver's.  The four-rows figure uses the tick of a stage that also holds embedding + lm_head (an upper bound on the stage time)"}
print(json.dumps(out))
