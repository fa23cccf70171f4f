#!/usr/bin/env python
"""bench.py — decode tokens/s of LLaMA-7B fp32 on MI355X, as a fraction of the per-token HBM-read roofline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run with one
rank per GPU.  W untimed warm-up steps, then exactly K steps timed between barrier + torch.cuda.synchronize() pairs, MAX over
ranks, rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): LLaMA-7B shape (d 4096, 32 heads, 32 layers, ff 11008, vocab 32000), fp32,
random-init weights from the counter-based generator (seed 1234, generated directly in HBM), --context 128, prompt = 8 fixed
token ids evaluated as one prefill, then greedy decode at positions P = 8, 9, ...  A "step" is one pass of the hot path
(llama.Eval with N = 1: 32 x [rmsnorm, wq|wk|wv, rope, attention, wo, rmsnorm, w1|w3, silu*mul, w2] + lm_head) per stream.

N = 1: one stream on the CONTRACT ROUTE: every step is one llama.Eval of one token = one ml_GraphCompute = one
       lh_graph_compute (the drop-in boundary; the fused plan replays a captured hipGraph inside it) + the host argmax of its logits row,
       driven by the host library's own loop (llamago_GreedyContinue).  `value` = steps / the time between HIP events recorded on the stream
       when each lh_graph_compute starts and behind its last launch (lh_ctx_time_computes), the measurement SURVEY §8d names for this config;
       the same K steps' whole-loop wall rate (host graph build, flatten, match, logits read, argmax included) stands beside it as
       "eval_per_token_loop", and the device-resident loop (argmax on the GPU feeds the next step, no host round trip: rounds 1-5's
       `value`, an extension of the boundary) as "resident_loop".
N > 1: layers are sharded in contiguous blocks over the ranks.  `value` = N independent greedy streams (server.go:88-101:
       the reference's --pods), ONE ROW PER WEIGHT PASS: N groups of one stream keep every rank busy every tick, a tick is the
       batch-1 decode step of N = 1 on the rank's L / N layers, and every token reads every weight exactly once - exactly like the
       N = 1 line, so value(N) / (N x value(1)) is the layer shard's efficiency with no footnote (scaling: weak; per-GPU work per
       step constant).  The residual row [4096 f32] hops rank r -> r+1 and the token ids return from the last rank to rank 0 as RCCL
       send/recv issued by the library itself (lh_pipeline_run: schedule, stages and p2p all below the C-ABI); torch.distributed
       (gloo) only carries the control plane: the 128-byte RCCL id, the barriers around the timed regions and the max-over-ranks
       of the times.  Side objects of the SAME invocation: "pods_batched_4n" (4 N streams sharing 4-row passes: rounds 3-5's
       headline - more tokens per weight byte, not comparable with N = 1's single stream), "single_stream" (one stream walking
       through the stages: the latency curve of SURVEY §8e) and "parity" (stream 0's ids against the committed single-GPU ids,
       tests/golden/7b_seed1234_ids.json).  `--pods P` / `--rows-per-pass M` override the first phase.  Started without torch.distributed.run, `--gpus N` spawns its own N ranks.  A rank
       that fails prints {"error": ...} and exits non-zero; control-plane waits time out after 120 s.

Extra objects on the JSON line: "roofline" (dominant kernel, HIP-event timed live), "cpu_baseline" (oracle on the host cores,
N = 1 only), "parity" (token ids / logits vs that oracle run).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL between ranks needs on this driver (must be set before the first HIP call)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s measured copy ceiling)
SEED = 1234


def bytes_per_token(hp_d, L, F, V, T):
    """weights touched once + KV read of T cached positions + KV write of one."""
    weights = 4 * (L * (4 * hp_d * hp_d + 3 * hp_d * F + 2 * hp_d) + V * hp_d + 2 * hp_d)
    kv = 2 * L * T * hp_d * 4 + 2 * L * hp_d * 4
    return weights, kv


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box reports 256
    logical CPUs but runs the job under a 16-CPU quota; oversubscribing OpenMP there is catastrophic)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU), relay rank 0's JSON line."""
    import socket
    import subprocess
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    # a rank that dies takes its siblings with it (exactly the PIDs started here): nobody waits for a peer that is gone
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for i, pr in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = pr.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    pr.kill()
                    rcs[i] = pr.wait()
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


_OUT = None   # the real stdout once main() has pointed fd 1 at stderr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--shape", default="7B")
    ap.add_argument("--layers", type=int, default=0, help="override layer count (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=0, help="decode steps of the CPU baseline / parity legs (default: all --steps, BASELINE.md §3)")
    ap.add_argument("--pods", type=int, default=0, help="independent greedy streams in flight for N > 1 (default N, one row per weight pass; 1 = the single-stream latency curve)")
    ap.add_argument("--rows-per-pass", type=int, default=1, help="N > 1: streams a tick evaluates in ONE pass over a rank's weights (default 1 = comparable with N = 1; 0 = as many as fit)")
    ap.add_argument("--no-prefill", action="store_true", help="skip the config-3 side measurement (13B, one 1024-token Eval)")
    ap.add_argument("--int8", action="store_true", help="BASELINE config 4: block-int8 weight matrices (36 B per 32 weights); not the headline metric")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    # stdout carries ONE JSON line.  Native libraries write there too ("[Gloo] Rank 0 is connected to 2 peer ranks" in front of the line of a three-rank
    # run: gloo's C++ side): fd 1 goes to stderr for the rest of the run, the line goes to the real stdout kept here
    global _OUT
    sys.stdout.flush()
    _OUT = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    ncpu_early = usable_cores()   # BEFORE an OpenMP runtime pins this thread (OMP_PROC_BIND below): the affinity mask then shows one core
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")  # before any OpenMP runtime loads (CPU baseline threads)
    # pinned threads (the CPU figure swung 4.9 -> 8.6 tok/s box to box with migrating ones), SPREAD over the sockets' cores: 16 neighbouring
    # cores share a few memory channels and gave 3.3 tok/s, stable but not what unpinned goroutines on an otherwise idle host get
    os.environ.setdefault("OMP_PROC_BIND", "spread")
    os.environ.setdefault("OMP_PLACES", "cores")
    import numpy as np
    import torch  # first: the process then uses ONE HIP runtime (torch's), libllamahip binds to it by SONAME
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # BENCH_SHARED_GPU=1: functional test of the N > 1 path on a box with ONE GPU (all ranks on device 0, p2p staged through
    # the host over gloo by the library's hook transport).  Never a measurement: RCCL cannot place two ranks on one device.
    shared_gpu = os.environ.get("BENCH_SHARED_GPU") == "1"
    if shared_gpu or os.environ.get("BENCH_SHARED_GPU") == "2":   # (2: all ranks on device 0 WITHOUT choosing the host-staged transport - RCCL refuses the
        local_rank = 0                                             #  communicator and the line must come out of the fallback below: its rehearsal)
    os.environ["LLAMAGO_DEVICE"] = str(local_rank)
    torch.cuda.set_device(local_rank)
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only; the data path is RCCL inside libllamahip.so.  120 s: a rank that crashed must not hold the others for gloo's default half hour
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=int(os.environ.get("BENCH_CONTROL_TIMEOUT_S", "120"))))

    import __graft_entry__ as graft
    if rank == 0:
        graft.build()
    if world > 1:
        dist.barrier()
    from llama_go_amd.mlapi import (MLLib, PROMPT, SHAPES, decode_greedy_resident, load_product, make_hparams, profile_decode)
    prod = load_product()

    kw = dict(SHAPES[args.shape])
    if args.layers:
        kw["layers"] = args.layers
    K, W = args.steps, args.warmup
    ctx_size = max(128, len(PROMPT) + K + W + 1 + 8)   # (+ 8: the head steps of the per-token host loop measurement)
    hp = make_hparams(**kw, ctx=ctx_size)
    d, L, V = hp.embdSize, hp.layersCount, hp.vocabSize
    PROMPT = [t % V for t in PROMPT]  # the fixed ids are 7B-vocabulary ids; debug shapes have smaller tables
    P0 = len(PROMPT)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    result = {}
    if world == 1:
        model = prod.NewSyntheticModel(hp, SEED)
        if args.int8:
            model.QuantizeQ8()
        F = model.ffSize
        wbytes_f32 = bytes_per_token(d, L, F, V, 0)[0]
        ctx = model.NewContext(ctx_size, 1)
        logits0 = ctx.Eval(PROMPT, 0)  # prefill through ml.GraphCompute (fused plan)
        first = int(np.argmax(logits0))
        # ---- the timed region: K steps of the contract route from the state behind the prompt (W warm-up steps from the same state first: they also
        # capture the step's hipGraph).  One step = llama.Eval([id], past) -> ml_GraphCompute -> lh_graph_compute, then the host argmax.
        if W > 0:
            ctx.GreedyContinue(first, P0, W)
        ctx.TimeComputes(True)
        sync_all()
        t0 = time.perf_counter()
        toks = ctx.GreedyContinue(first, P0, K)
        sync_all()
        dt_wall = time.perf_counter() - t0
        st_timed = ctx.ComputeStats()
        ctx.TimeComputes(False)
        last_logits = ctx.logits()
        assert st_timed["calls"] == K, st_timed
        dt = st_timed["device_us"] * 1e-6      # HIP events around each of the K lh_graph_compute calls, summed
        tokens_total = K
        tokens = [first] + toks[:-1]  # ids evaluated by the timed steps; toks = ids they produced
        produced = toks
        result["eval_per_token_loop"] = {"tokens_per_s": round(K / dt_wall, 2), "ms_per_token": round(dt_wall / K * 1e3, 4),
                                         "inside_lh_graph_compute_wall": {"tokens_per_s": round(K / max(st_timed["wall_us"], 1e-9) * 1e6, 2), "ms_per_call": round(st_timed["wall_us"] / K / 1e3, 4),
                                                                          "note": "host clock from entry to return of the contract call (validate, match, enqueue, wait, pinned logits row)"},
                                         "note": "the SAME K timed steps by the wall clock between the barrier + synchronize pairs: llama.Eval per token - the host mirror moves the graph it keeps for one-token Evals to the new position (1 us; "
                                                 "building it anew as the reference's llama.Eval does cost 53 + 13 us, rounds 1-5) - the library's match (12 us), the wait, the logits row and the host argmax; "
                                                 "`value` is the HIP-event time of the lh_graph_compute calls inside it (it starts at the call's entry: validate and match are in it)"}
        # ---- rounds 1-5's `value`: the device-resident loop (argmax on the GPU feeds the next step, hipGraph replay of eight steps, no host round trip)
        cr = model.NewContext(ctx_size, 1)
        cr.Eval(PROMPT, 0)
        if W > 0:
            decode_greedy_resident(cr, first, P0, W)
        torch.cuda.synchronize()
        t_r = time.perf_counter()
        toks_res, _ = decode_greedy_resident(cr, first, P0, K, want_logits=True)
        torch.cuda.synchronize()
        dt_res = time.perf_counter() - t_r
        cr.free()
        result["resident_loop"] = {"tokens_per_s": round(K / dt_res, 2), "ms_per_token": round(dt_res / K * 1e3, 4), "ids_equal_contract_route": [int(t) for t in toks_res] == [int(t) for t in toks],
                                   "note": "lh_llama_decode_greedy: the whole loop on the device (an extension of the boundary, not the contract call); `value` of rounds 1-5"}
        # ---- the reference's real generation loop: SampleTopPTopK (topK 40, topP 0.95, repeat penalty 1.10 — main.go:87-90) after every
        # Eval, sampler resident on the device (no logits D2H).  Decode rate = (t[K+1 samples] - t[1 sample]) / K, same prefill in both.
        SMP = dict(topK=min(40, V), topP=0.95, temp=0.8, repeatPenalty=1.10, seed=SEED)
        c4 = model.NewContext(ctx_size, 1)
        c4.SampleDecode(PROMPT, min(W, 2) + 1, **SMP)
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        c4.SampleDecode(PROMPT, 1, **SMP)
        t_s1 = time.perf_counter()
        sampled = c4.SampleDecode(PROMPT, K + 1, **SMP)
        t_s2 = time.perf_counter()
        c4.free()
        smp_dt = max(1e-9, (t_s2 - t_s1) - (t_s1 - t_s))
        result["sampled_decode"] = {"tokens_per_s": round(K / smp_dt, 2), "ms_per_token": round(smp_dt / K * 1e3, 4), "params": SMP,
                                    "note": "llama.SampleTopPTopK on the device after every Eval (server.go:201-204); counter-based uniforms (not `value`)"}
        # ---- dominant kernel, HIP-event timed with eager launches of the same kernels (all weights distinct: HBM-cold)
        prof = profile_decode(ctx, first, P0, repeats=2)
        result["kernels"] = {k["name"]: {"avg_us": round(k["avg_us"], 2), "launches": k["launches"], "GBps": round(k["gbps"], 1)} for k in prof if not k["name"].endswith("/b2b")}
        b2b = {k["name"][:-4]: k for k in prof if k["name"].endswith("/b2b")}
        prof = [k for k in prof if not k["name"].endswith("/b2b")]
        dom = max(prof, key=lambda k: k["avg_us"] * k["launches"])
        per_pair_us = dom["avg_us"]
        if dom["name"] in b2b:  # the dominant kernel's launches of a step back to back between ONE event pair (no per-kernel event cost)
            dom = dict(b2b[dom["name"]], name=dom["name"])
        traffic, traffic_src, traffic_ok = None, None, None
        try:  # HBM bytes per launch from the committed PMC pass (counters cannot be collected from inside this process); only believed while the
            # kernel sources are the ones the pass profiled (tools/source_hash.py: sha256 over csrc/ + include/llamahip.h, stamped by tools/pmc_summary.py)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from source_hash import kernel_source_hash
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            if not args.int8 and args.shape == "7B":
                traffic_ok = pmc.get("kernel_source_sha256") == kernel_source_hash()
                if traffic_ok:
                    traffic, traffic_src = pmc["bytes_per_launch"].get(dom["name"]), pmc["source"]
                else:
                    traffic_src = ("dropped: profiles/pmc_traffic.json was measured on other kernel sources (kernel_source_sha256 differs) - rerun the pmc:f32 step of "
                                   "tools/gpu/r6_final.sh")
        except Exception:
            pass
        # what this box's HBM delivers to a kernel that only reads (lh_hbm_read_probe: the weight stream's access pattern, no arithmetic),
        # measured live: the nominal 8 TB/s is not reachable by any access pattern; SURVEY 8d asks for both yardsticks
        meas = None
        try:
            from llama_go_amd.mlapi import hbm_read_probe
            meas = hbm_read_probe(prod)
        except Exception:
            pass
        result["roofline"] = {
            "bound": "hbm", "kernel": dom["name"], "achieved": round(dom["gbps"], 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(dom["gbps"] / HBM_PEAK_GBPS, 4), "measured_read_stream_GBps": round(meas, 1) if meas else None,
            "frac_of_measured_stream": round(dom["gbps"] / meas, 4) if meas else None, "traffic": traffic, "traffic_source": traffic_src, "traffic_build_matches": traffic_ok,
            "bytes_per_launch": dom["bytes_per_launch"], "avg_us": round(dom["avg_us"], 2), "avg_us_with_event_pair_per_launch": round(per_pair_us, 2),
            "note": "algorithmic bytes = rows*cols*4 of the weights one launch streams (SURVEY 8d); avg_us = HIP events around the kernel's 32 launches "
                    "of a step, back to back, all weights distinct (agrees with the rocprofv3 kernel trace in profiles/); traffic (PMC) in profiles/",
        }
        # ---- CPU baseline + parity on the same inputs (oracle = test infrastructure; only used here as the checker / baseline)
        if not args.no_cpu_baseline:
            orc = MLLib(os.path.join(ROOT, "oracle", "liboracle.so"))
            ncpu = ncpu_early
            t_gen = time.perf_counter()
            om = orc.NewSyntheticModel(hp, SEED)
            if args.int8:
                om.QuantizeQ8()
            t_gen = time.perf_counter() - t_gen
            nsteps = max(1, min(args.cpu_steps or K, K))
            # (B) "--avx-equivalent": the reference's own vdot (oracle/_ref), rows chunked over the host cores as ml.go:2010-2013 does;
            # three samples of the same nsteps (fresh context each), the MEDIAN is reported
            avx_samples = []
            for rep_ in range(3):
                oc = om.NewContext(ctx_size, ncpu, True)
                lg = oc.Eval(PROMPT, 0)
                tok = int(np.argmax(lg))
                avx_logits = [lg]
                t1 = time.perf_counter()
                for s in range(nsteps):
                    lg = oc.Eval([tok], P0 + s)
                    avx_logits.append(lg)
                    tok = int(np.argmax(lg))
                avx_samples.append((time.perf_counter() - t1) / nsteps)
                oc.free()
            avx_dt = sorted(avx_samples)[1]
            # (A) "pure-Go-equivalent" scalar order = the PARITY reference: all cores only redistribute rows (same values)
            oc = om.NewContext(ctx_size, ncpu, False)
            otoks, ologits = oc.GreedyDecode(PROMPT, nsteps + 1)
            oc.free()
            # (C) float64-accumulation truth leg: the yardstick for (A), (B) and the GPU's tree order
            oc = om.NewContext(ctx_size, ncpu, 2)
            ftoks, flogits = oc.GreedyDecode(PROMPT, nsteps + 1)
            oc.free()
            oc = om.NewContext(ctx_size, ncpu, False)
            osampled = oc.SampleDecode(PROMPT, min(nsteps, 4) + 1, **SMP)
            oc.free()
            # one scalar step on ONE thread for the 1-thread figure of BASELINE.md row A
            oc1 = om.NewContext(ctx_size, 1, False)
            t2 = time.perf_counter()
            oc1.Eval([7], 0)
            scalar1_dt = time.perf_counter() - t2
            oc1.free()
            om.free()
            result["cpu_baseline"] = {
                "value": round(1.0 / avx_dt, 3), "unit": "tokens/s", "cores": ncpu, "kind": "port",
                "sample": f"checker's restatement of the Eval schedule with the reference's row chunking per MulMat (ml.go:2010-2013: thread i of n takes "
                          f"rows [i*ceil(nr/n), ...), OpenMP threads pinned to cores standing in for the goroutines) calling the reference's own "
                          f"utils/floats_avx.c vdot (oracle/_ref) = '--avx' path; full {args.shape} model, {nsteps} decode steps at P={P0}.. after an {P0}-token "
                          f"prefill, {ncpu} host threads; median of 3 samples",
                "ms_per_token": round(avx_dt * 1e3, 1), "samples_tokens_per_s": [round(1.0 / t, 3) for t in avx_samples],
                "pure_go_scalar_1thread_ms_per_token": round(scalar1_dt * 1e3, 1),
                "weights_gen_s": round(t_gen, 1), "host_logical_cpus": os.cpu_count(),
            }
            # parity: GPU ids/logits (llama.Eval per token through ml_GraphCompute) vs the scalar-order checker on ALL compared steps
            c2 = model.NewContext(ctx_size, 1)
            c2.Eval(PROMPT, 0)
            gt, glog = [first], []
            for s in range(nsteps):
                l2 = c2.Eval([gt[-1]], P0 + s)
                glog.append(l2)
                gt.append(int(np.argmax(l2)))
            c2.free()
            gl_all = np.stack([logits0] + glog)

            def relerr(a, b):  # max over steps of max|a-b| / max|b|
                return float((np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max(axis=-1) / np.abs(b).max(axis=-1)).max())

            srt = np.sort(ologits, axis=-1)
            result["parity"] = {
                "token_ids_match": gt[: nsteps + 1] == list(otoks[: nsteps + 1]) and gt[1: nsteps + 1] == produced[:nsteps],
                "max_rel_logit_err": relerr(gl_all, ologits), "tolerance": 1e-4, "steps_compared": nsteps + 1,
                "min_top2_margin_rel": float(((srt[:, -1] - srt[:, -2]) / np.abs(ologits).max(axis=-1)).min()),
                "sampled_token_ids_match": list(sampled[: len(osampled)]) == list(osampled),
                "f64_leg": {"token_ids_match_gpu": gt[: nsteps + 1] == list(ftoks), "gpu_vs_f64": relerr(gl_all, flogits),
                            "scalar_go_order_vs_f64": relerr(ologits, flogits), "avx_order_vs_f64": relerr(np.stack(avx_logits), flogits),
                            "note": "float64-accumulated dot products, one rounding (checker mode useAVX=2); max over steps of max|delta|/max|truth|"},
            }
        # ---- the reference's --pods on ONE GPU (server.go:88-101: MaxPods concurrent Do() over one Model): P streams whose decode
        # steps share one pass over the weights (lh_batch through the pipeline scheduler, world = 1) - aggregate tokens/s, not `value`
        if not args.no_prefill:
            try:
                from llama_go_amd.mlapi import Pipeline
                pb = {}
                wb_tick, kv_row = bytes_per_token(d, L, F, V, P0 + W + (K + 1) / 2.0)
                macs_row = L * (4 * d * d + 3 * d * F) + V * d   # weight MACs of one stream's token
                for P in (4, 8, 16, 32, 64):
                    plb = Pipeline(model, ctx_size, P, 0, 1)
                    plb.run([PROMPT] * P, max(W, 1))
                    torch.cuda.synchronize()
                    t_b = time.perf_counter()
                    plb.run(None, K)
                    torch.cuda.synchronize()
                    d_b = time.perf_counter() - t_b
                    ids_b = [plb.tokens(i) for i in range(P)]
                    plb.free()
                    solo = [first] + toks   # the single stream's ids from the same prompt (timed region above)
                    n_c = min(len(solo), len(ids_b[0]))
                    tick = d_b / K
                    # a tick reads the weights once + every row's KV; its matmuls are P rows deep: both rooflines, the binding one named
                    f_hbm = (wb_tick + P * kv_row) / tick / (HBM_PEAK_GBPS * 1e9)
                    f_mfma = 2.0 * P * macs_row / tick / 157.3e12
                    pb[str(P)] = {"tokens_per_s": round(P * K / d_b, 1), "ms_per_tick": round(tick * 1e3, 4),
                                  "frac_of_hbm_roofline": round(f_hbm, 4), "frac_of_fp32_mfma_peak": round(f_mfma, 4), "bound": "hbm" if f_hbm >= f_mfma else "mfma",
                                  "ids_match_single_stream": all(t[:n_c] == solo[:n_c] for t in ids_b)}
                    if P >= 49 and not args.int8:   # 49..64 rows: k_stream_b9 - eight exact bf16 products per weight (round 6), priced against ITS pipe as well
                        pb[str(P)]["matmul_kernel"] = "k_stream_b9 (bf16 matrix pipe, 8 exact products per weight)"
                        pb[str(P)]["frac_of_exact_split_bf16_roof"] = round(2.0 * P * macs_row * 8 / tick / 2.5e15, 4)
                result["pods_batched"] = dict(pb, note="P independent greedy streams on ONE GPU, one pass over the weights per tick for all of them "
                                                       "(rows = pods: the decode stream itself up to 8 rows, the MFMA stream kernels beyond; per-row KV cache and position); "
                                                       "aggregate tokens/s; fractions against 8 TB/s and the 157.3 TFLOP/s fp32 matrix peak (the chip holds ~2.09 GHz under the fp32-MFMA "
                                                       "stream kernels, i.e. ~137 TFLOP/s: profiles/r04_stream_eight_tiles_clock.txt); from 49 pods the matmuls run on the bf16 pipe as eight exact "
                                                       "products per weight (k_stream_b9): frac_of_exact_split_bf16_roof = 8 x the fp32-equivalent flops against 2.5 PFLOP/s - the kernel is "
                                                       "POWER-bound there (1.6-1.9 GHz, profiles/r06_stream_b9_probe.txt)")
            except Exception as e:  # a side measurement must never take the headline line down
                result["pods_batched"] = {"error": str(e)}
        ctx.free()
        model.free()
        # ---- BASELINE config 3 beside the headline (not `value`): LLaMA-13B fp32, ONE Eval of 1024 tokens at past = 0 -> TFLOP/s on
        # the executed weight-matmul flops against the 157.3 TFLOP/s fp32 MFMA peak (tools/bench_prefill.py is the standalone form)
        if not args.no_prefill and args.shape == "7B" and not args.layers:
            try:
                hp13 = make_hparams(**SHAPES["13B"], ctx=1024)
                m13 = prod.NewSyntheticModel(hp13, SEED)
                if args.int8:
                    m13.QuantizeQ8()
                c13 = m13.NewContext(1024, 1)
                toks13 = [int(t) for t in np.random.default_rng(0).integers(0, hp13.vocabSize, 1024)]
                c13.Eval(toks13, 0)
                ts13 = []
                for _ in range(2):
                    torch.cuda.synchronize()
                    t13 = time.perf_counter()
                    c13.Eval(toks13, 0)
                    ts13.append(time.perf_counter() - t13)
                d13, L13, F13 = hp13.embdSize, hp13.layersCount, m13.ffSize
                fl13 = 2.0 * 1024 * L13 * (4 * d13 * d13 + 3 * d13 * F13) + 2.0 * hp13.vocabSize * d13
                result["prefill_13b"] = {"seconds": round(min(ts13), 4), "n_tokens": 1024, "tflop_weight_matmuls_executed": round(fl13 / 1e12, 2),
                                         "TFLOPs_per_s": round(fl13 / min(ts13) / 1e12, 1), "frac_of_fp32_mfma_peak_157.3": round(fl13 / min(ts13) / 157.3e12, 3),
                                         "frac_of_exact_split_bf16_roof_312.5": round(fl13 * 8 / min(ts13) / 2.5e15, 3),
                                         "roofs": "the weight GEMMs run on the bf16 matrix pipe as EIGHT exact products per fp32 product (k_gemm_b9: xl * wl dropped, error vs f64 unchanged - "
                                                  "profiles/r06_gemm_b9_products.txt): 157.3 TF = the fp32-input MFMA peak it replaces, 312.5 TF = 2.5 PF bf16 / 8 = the roof of the exact split "
                                                  "itself; the kernel is POWER-bound (1.4-1.7 GHz with every CU on the bf16 pipe), not issue-bound",
                                         "note": "llama.Eval of 1024 tokens incl. host graph build and last-row logits D2H; lm_head for the one row Eval reads"}
                c13.free()
                if not args.int8:   # the same Eval on the same model with block-int8 weight matrices (config 4's format at config 3's size; k_gemm_q8b3)
                    m13.QuantizeQ8()
                    c13 = m13.NewContext(1024, 1)
                    c13.Eval(toks13, 0)
                    tq13 = []
                    for _ in range(2):
                        torch.cuda.synchronize()
                        t13 = time.perf_counter()
                        c13.Eval(toks13, 0)
                        tq13.append(time.perf_counter() - t13)
                    result["prefill_13b"]["block_int8_seconds"] = round(min(tq13), 4)
                    c13.free()
                m13.free()
            except Exception as e:  # a side measurement must never take the headline line down
                result["prefill_13b"] = {"error": str(e)}
        # ---- side objects the driver should see on the default line (not `value`): time to first token for the 8-token prompt, and
        # BASELINE config 4 (block-int8 weights) through the same resident decode loop
        if args.shape == "7B" and not args.layers and not args.int8 and not args.no_prefill:
            try:
                mt = prod.NewSyntheticModel(hp, SEED)
                ct = mt.NewContext(ctx_size, 1)
                ct.Eval(PROMPT, 0)
                tt = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t_p = time.perf_counter()
                    ct.Eval(PROMPT, 0)
                    tt.append(time.perf_counter() - t_p)
                result["prompt_8_tokens"] = {"ms": round(min(tt) * 1e3, 3), "floor_ms_weight_stream_at_8TBps": round(wbytes_f32 / 8e12 * 1e3, 2),
                                             "note": "one llama.Eval of the 8-token prompt at past = 0 (server.go:185-192), host graph build + logits D2H included"}
                by_len = {}
                for n_p in (16, 32, 64):   # longer prompts of the same kind (ids cycled from the fixed prompt): the streaming MFMA kernel up to 32 rows, the tile GEMM above
                    toks = [PROMPT[i % len(PROMPT)] for i in range(n_p)]
                    ct.Eval(toks, 0)
                    tl = []
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t_p = time.perf_counter()
                        ct.Eval(toks, 0)
                        tl.append(time.perf_counter() - t_p)
                    by_len[str(n_p)] = round(min(tl) * 1e3, 3)
                ct.free()
                ct = mt.NewContext(256, 1)   # (the decode phases' window is shorter than these prompts)
                for n_p in (128, 160):       # 128: one pass of the stream kernels; 160: every matrix in two passes of 80 rows (round 6; the tile GEMM before)
                    toks = [PROMPT[i % len(PROMPT)] for i in range(n_p)]
                    ct.Eval(toks, 0)
                    tl = []
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t_p = time.perf_counter()
                        ct.Eval(toks, 0)
                        tl.append(time.perf_counter() - t_p)
                    by_len[str(n_p)] = round(min(tl) * 1e3, 3)
                result["prompt_8_tokens"]["ms_by_prompt_length"] = by_len
                ct.free()
                mt.QuantizeQ8()
                cq = mt.NewContext(ctx_size, 1)
                fq = int(np.argmax(cq.Eval(PROMPT, 0)))
                decode_greedy_resident(cq, fq, P0, W or 1)
                torch.cuda.synchronize()
                t_q = time.perf_counter()
                tq, _ = decode_greedy_resident(cq, fq, P0, K)
                torch.cuda.synchronize()
                dq = time.perf_counter() - t_q
                mat = 4 * (L * (4 * d * d + 3 * d * F) + V * d)
                qbytes = (wbytes_f32 - mat) + mat * 36 // 128
                result["int8_decode"] = {"tokens_per_s": round(K / dq, 2), "ms_per_token": round(dq / K * 1e3, 4), "bytes_per_token": int(qbytes),
                                         "frac_of_hbm_roofline": round(K / dq * qbytes / (HBM_PEAK_GBPS * 1e9), 4), "tokens": tq[: min(K, 16)],
                                         "note": "BASELINE config 4: block-int8 weight matrices (36 B per 32 weights, format ours), same resident loop"}
                # what a chain of dependent launches can reach: every launch costs t = 3.2 us + bytes / 7.0 TB/s (fit over the five fp32 decode launches of a
                # layer; the five per layer are one all-to-all dependency chain), and the int8 streams are 3.55x shorter than the fp32 ones
                n_launch = 5 * L + 2
                chain_s = n_launch * 3.2e-6 + qbytes / 7.0e12
                result["int8_decode"]["launch_chain_ceiling_frac"] = round(qbytes / chain_s / (HBM_PEAK_GBPS * 1e9), 4)
                result["int8_decode"]["launch_chain_ceiling"] = {"tokens_per_s": round(1.0 / chain_s, 1), "launches_per_token": n_launch,
                                                                 "fit": "t(launch) = 3.2 us + bytes / 7.0 TB/s, five dependent launches per layer + lm_head + argmax (DESIGN 3a)",
                                                                 "note": "structural: 558 -> 550 -> 541 tok/s over rounds 3-5 (resident kernels, cross-kernel prefetch, two-stream overlap, launch-free "
                                                                         "attention and the Infinity-Cache prefetch all measured, all lost: DESIGN 3a); round 6 took scalar work out of "
                                                                         "the launches' first microsecond (row addressing of wq|wk|wv, the row-block divisions, flat cache loads in the "
                                                                         "attention): 551 -> 570 tok/s same box (profiles/r06_q8s_phase_probe.txt)"}
                # dominant int8 kernel, HIP-event timed like the fp32 one (bytes = 36 B per 32 weights of the launch)
                pq = profile_decode(cq, fq, P0, repeats=2)
                b2q = {k["name"][:-4]: k for k in pq if k["name"].endswith("/b2b")}
                pq = [k for k in pq if not k["name"].endswith("/b2b")]
                dq_k = max(pq, key=lambda k: k["avg_us"] * k["launches"])
                if dq_k["name"] in b2q:
                    dq_k = dict(b2q[dq_k["name"]], name=dq_k["name"])
                result["int8_decode"]["roofline"] = {"bound": "hbm", "kernel": dq_k["name"], "achieved": round(dq_k["gbps"], 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                     "frac": round(dq_k["gbps"] / HBM_PEAK_GBPS, 4), "bytes_per_launch": dq_k["bytes_per_launch"], "avg_us": round(dq_k["avg_us"], 2)}
                result["int8_decode"]["kernels"] = {k["name"]: {"avg_us": round(k["avg_us"], 2), "GBps": round(k["gbps"], 1)} for k in pq}
                # the int8 pods of one GPU in one weight pass (k_gemv_q8_rows up to 4 rows, k_stream_q8 beyond: raw bytes by LDS-DMA, dequantised by the
                # MFMA waves) and the 8-token prompt, beside their fp32 twins above
                try:
                    from llama_go_amd.mlapi import Pipeline
                    pq8 = {}
                    for P in (4, 8, 16, 32):
                        plq = Pipeline(mt, ctx_size, P, 0, 1)
                        plq.run([PROMPT] * P, max(W, 1))
                        torch.cuda.synchronize()
                        t_b = time.perf_counter()
                        plq.run(None, K)
                        torch.cuda.synchronize()
                        d_b = time.perf_counter() - t_b
                        idq = [plq.tokens(i) for i in range(P)]
                        plq.free()
                        pq8[str(P)] = {"tokens_per_s": round(P * K / d_b, 1), "ms_per_tick": round(d_b / K * 1e3, 4), "all_streams_equal": all(t == idq[0] for t in idq)}
                    result["int8_decode"]["pods_batched"] = pq8
                    cq.Eval(PROMPT, 0)
                    tq8 = []
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t_p = time.perf_counter()
                        cq.Eval(PROMPT, 0)
                        tq8.append(time.perf_counter() - t_p)
                    result["int8_decode"]["prompt_8_tokens_ms"] = round(min(tq8) * 1e3, 3)
                except Exception as e:
                    result["int8_decode"]["pods_batched"] = {"error": str(e)}
                # the ids of ALL timed steps + the last logits against the dequantise-then-fp32 checker on the full model (scalar Go order)
                lgq = None
                if not args.no_cpu_baseline:
                    cq2 = mt.NewContext(ctx_size, 1)
                    glq = [cq2.Eval(PROMPT, 0)]
                    for s_ in range(K):
                        glq.append(cq2.Eval([int(np.argmax(glq[-1]))], P0 + s_))
                    cq2.free()
                    gq_ids = [int(np.argmax(l_)) for l_ in glq]
                    orc_q = MLLib(os.path.join(ROOT, "oracle", "liboracle.so"))
                    omq = orc_q.NewSyntheticModel(hp, SEED)
                    omq.QuantizeQ8()
                    ocq = omq.NewContext(ctx_size, ncpu_early, False)
                    oq_ids, oq_lg = ocq.GreedyDecode(PROMPT, K + 1)
                    ocq.free()
                    omq.free()
                    errq = float((np.abs(np.stack(glq).astype(np.float64) - oq_lg).max(axis=-1) / np.abs(oq_lg).max(axis=-1)).max())
                    result["int8_decode"]["parity"] = {"ids_match": gq_ids == list(oq_ids) and gq_ids[1:] == tq[:K], "max_rel_logit_err": errq, "tolerance": 1e-4,
                                                       "steps_compared": K + 1, "checker": "dequantise to fp32 (fl32(d*q)), then the fp32 Eval in scalar Go order"}
                cq.free()
                mt.free()
            except Exception as e:  # a side measurement must never take the headline line down
                result["int8_decode"] = {"error": str(e)}
        parallelism = "single GPU, one lh_graph_compute per token (fused plan, hipGraph replay inside the call), host argmax"
        pods = 1
        groups = 1
        timed_pos0 = P0
    else:
        # ---------------- layer-sharded pipeline over `world` ranks ----------------
        R = world
        # streams in flight: R groups of ONE keep every rank busy every tick, each tick the batch-1 decode step on the rank's layers (every
        # token reads every weight once, as on one GPU); a phase of K steps costs K + (R - 1) ticks per rank (pipeline fill + drain
        # included in the timed region).  The reference's own knob for the stream count is --pods (server.go:88-101).
        pods = args.pods or R
        rows_per_pass_arg = args.rows_per_pass
        from llama_go_amd.mlapi import Pipeline, comm_unique_id
        from llama_go_amd.pipeline import gloo_comm_hooks, layer_range
        l0, l1 = layer_range(rank, R, L)

        def fail(stage, e):   # a failing rank says so and leaves: the launcher (or spawn_ranks) takes the others down
            msg = {"error": f"rank {rank}: {stage}: {e}", "n_gpus": world}
            print(json.dumps(msg), file=sys.stderr if rank else _OUT, flush=True)
            os._exit(1)

        try:
            model = prod.NewSyntheticModel(hp, SEED, l0, l1)
            if args.int8:
                model.QuantizeQ8()
        except Exception as e:
            fail("model", e)
        F = model.ffSize

        transport = {"kind": "host-staged over gloo (BENCH_SHARED_GPU)" if shared_gpu else "rccl"}

        def new_pipeline(n_streams, max_rows=0):
            if transport["kind"] == "rccl":
                obj = [comm_unique_id(prod) if rank == 0 else None]   # ncclGetUniqueId on rank 0; any channel may carry the 128 bytes
                dist.broadcast_object_list(obj, src=0)
                pl, err = None, None
                try:
                    pl = Pipeline(model, ctx_size, n_streams, rank, R, comm_id=obj[0], max_rows=max_rows)   # ncclCommInitRank + per-stream stages, one HIP stream
                except Exception as e:
                    err = f"rank {rank}: {e}"
                errs = [None] * world
                dist.all_gather_object(errs, err)
                errs = [e for e in errs if e]
                if not errs:
                    return pl
                # RCCL with N > 1 ranks has never run on this code's development boxes (one GPU each).  If a rank cannot join the communicator the line is
                # still worth having: every rank drops to the library's hook transport (residual rows staged through the host over the gloo control group)
                # and the line says so - a functional result with the transport's cost in it, not the xGMI number.
                if pl is not None:
                    pl.free()
                transport["kind"] = "host-staged over gloo (RCCL communicator failed: " + errs[0][:300] + ")"
                if rank == 0:
                    print(f"[bench] RCCL pipeline failed ({errs[0]}); falling back to the host-staged transport", file=sys.stderr, flush=True)
            return Pipeline(model, ctx_size, n_streams, rank, R, hooks=gloo_comm_hooks(dist), max_rows=max_rows)

        def timed_phase(n_streams, max_rows=0):
            pl = new_pipeline(n_streams, max_rows)
            pl.run([PROMPT] * n_streams, W)   # prefill + W warm-up decode steps per stream (lh_pipeline_run); the pipeline drains at the end
            sync_all()
            t0 = time.perf_counter()
            pl.run(None, K)                   # K decode steps per stream, continuing each stream: includes pipeline fill + drain
            sync_all()
            tdt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
            dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
            # ids rank 0 received for stream 0: [prefill argmax, then the id produced at positions P0, P0+1, ...]
            obj = [[pl.tokens(i) for i in range(n_streams)] if rank == 0 else None]
            dist.broadcast_object_list(obj, src=0)
            return pl, float(tdt.item()), obj[0]

        try:
            pl, dt, all_ids = timed_phase(pods, rows_per_pass_arg)
        except Exception as e:
            fail(f"pipeline phase with {pods} streams", e)
        tokens_total = K * pods
        groups = pl.groups
        prof = pl.profile_decode(1, P0, repeats=2)
        b2b = {k["name"][:-4]: k for k in prof if k["name"].endswith("/b2b")}
        prof = [k for k in prof if not k["name"].endswith("/b2b")]
        dom = max(prof, key=lambda k: k["avg_us"] * k["launches"])
        if dom["name"] in b2b:
            dom = dict(b2b[dom["name"]], name=dom["name"])
        result["roofline"] = {"bound": "hbm", "kernel": dom["name"], "achieved": round(dom["gbps"], 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round(dom["gbps"] / HBM_PEAK_GBPS, 4), "traffic": None, "bytes_per_launch": dom["bytes_per_launch"],
                              "avg_us": round(dom["avg_us"], 2),
                              "note": "the rank's batch-1 weight-stream kernel, HIP-event timed (the timed ticks carry several rows through the same stream: k_gemv_rows up to 4 rows per pass, the stream-GEMM kernels beyond)"}
        allt = all_ids[0]
        assert len(allt) == 1 + W + K, (len(allt), W, K)
        pl.free()
        # Side measurements from here on: a failure in one of them must not cost the headline line that is already measured.  A rank
        # that fails records the error; the ranks compare notes (gloo) behind every side phase and skip the remaining ones together, so nobody
        # walks alone into a collective.  (A rank that fails INSIDE a library collective aborts the communicator: its peers fail too, and say so.)
        def all_ranks_ok(err):
            flags = [None] * world
            dist.all_gather_object(flags, err)
            return [f for f in flags if f], flags

        side_alive = True
        # ---- rounds 3-5's headline as a side object: 4 N streams, a tick = ONE 4-row pass over the rank's weights (lh_batch)
        if pods == R and rows_per_pass_arg == 1:
            err = None
            try:
                pl4, dt4, ids4 = timed_phase(4 * R)
                result["pods_batched_4n"] = {"tokens_per_s": round(4 * R * K / dt4, 2), "ms_per_step": round(dt4 / K * 1e3, 4), "streams": 4 * R, "groups": pl4.groups,
                                             "rows_per_weight_pass": 4 * R // pl4.groups, "all_streams_equal_value_stream0": all(t == allt for t in ids4),
                                             "note": "more tokens per weight byte than `value`: compare with pods_batched['4'] of the N = 1 line, not with its single stream"}
                pl4.free()
            except Exception as e:
                err = f"rank {rank}: {e}"
            bad, _ = all_ranks_ok(err)
            if bad:
                result["pods_batched_4n"] = {"error": "; ".join(bad)}
                side_alive = False
        # ---- second curve of SURVEY §8e in the same invocation: ONE greedy stream walking through the stages (latency, not throughput)
        if pods != 1 and side_alive:
            err = None
            try:
                pl1, dt1, ids1 = timed_phase(1)
                pl1.free()
                result["single_stream"] = {"tokens_per_s": round(K / dt1, 2), "ms_per_token": round(dt1 / K * 1e3, 4), "steps": K,
                                           "ids_match_batched_stream0": ids1[0] == allt,
                                           "note": "pods = 1: every token passes the stages one after the other (R weight-stream stages + R hops per token); not `value`"}
            except Exception as e:
                err = f"rank {rank}: {e}"
            bad, _ = all_ranks_ok(err)
            if bad:
                result["single_stream"] = {"error": "; ".join(bad)}
                side_alive = False
        # ---- where the time goes, per rank (not part of the timed runs above: three event records per tick): the rank's own stage, the
        # exchange behind it (RCCL: includes waiting for the predecessor, i.e. pipeline bubbles), and a pre-flight of the ring itself
        if side_alive:
            mine, err = None, None
            try:
                pld = new_pipeline(pods, rows_per_pass_arg)
                pld.run([PROMPT] * pods, 2)
                hop_us = pld.hop_probe(d * 4, 1000)          # one residual row (16 KB on 7B), every rank shifting at the same time
                hop_us_tick = pld.hop_probe(d * 4 * max(1, pods // pld.groups), 200)   # the rows of one tick's exchange
                sync_all()
                pld.profile(True)
                kd = min(K, 8)
                pld.run(None, kd)
                st = pld.stats()
                pld.free()
                mine = {"rank": rank, "layers": l1 - l0, "ticks": st["ticks"], "stage_ms_per_tick": round(st["stage_ms"] / max(st["ticks"], 1), 4),
                        "exchange_us_per_tick": round(st["exchange_ms"] * 1e3 / max(st["ticks"], 1), 2), "hop_us_one_row": round(hop_us, 2), "hop_us_tick_rows": round(hop_us_tick, 2)}
            except Exception as e:
                err = f"rank {rank}: {e}"
            bad, _ = all_ranks_ok(err)
            if bad:
                result["pipeline_breakdown"] = {"error": "; ".join(bad)}
            else:
                allst = [None] * world
                dist.all_gather_object(allst, mine)
                result["pipeline_breakdown"] = {"by_rank": allst, "steps": min(K, 8), "streams": pods,
                                                "note": "HIP events on each rank's compute stream around every tick's stage and exchange (lh_pipeline_profile); exchange includes waiting for the "
                                                        "predecessor's rows; hop_us = grouped send + receive round the ring (lh_pipeline_hop_probe)"}
        else:
            result["pipeline_breakdown"] = {"error": "skipped: the single-stream phase failed"}
        # ---- parity: every stream saw the same prompt, so all of them must produce the ids of the single-GPU run (committed with the
        # checker's verdict on them: tests/golden/7b_seed1234_ids.json, written by tools/make_golden_ids.py from a bench.py --gpus 1 run)
        par = {"all_streams_equal": all(t == allt for t in all_ids), "ids_match_single_gpu": None}
        gpath = os.path.join(ROOT, "tests", "golden", "7b_seed1234_ids.json")
        if args.shape == "7B" and not args.layers and not args.int8 and os.path.exists(gpath):
            gold = json.load(open(gpath))
            n_cmp = min(len(allt), len(gold["ids"]))
            par.update(ids_match_single_gpu=allt[:n_cmp] == gold["ids"][:n_cmp], ids_compared=n_cmp, golden="tests/golden/7b_seed1234_ids.json",
                       golden_checked_against_oracle=gold.get("oracle_ids_match"))
        else:
            par["reason"] = "golden ids exist for the full fp32 7B model only"
        result["parity"] = par
        result["scaling_reference"] = {"ideal": f"{R} x value(N = 1)", "why": "N streams, one row per weight pass: each token reads every weight once on N GPUs as on one; "
                                       "value(N) / (N x value(1)) = efficiency of the layer shard (pipeline fill + drain of K + N - 1 ticks and the hops included)",
                                       "rows_per_weight_pass": max(1, pods // groups)}
        produced = allt[1:]          # ids produced by the decode steps at positions P0.. (the first W of them by the warm-up)
        timed_pos0 = P0 + W
        parallelism = (f"layer-shard pp{R} ({l1 - l0} layers on this rank), {pods} independent greedy stream{'s' if pods > 1 else ''} in flight as {groups} "
                       f"group{'s' if groups > 1 else ''} of {pods // groups} (one pass over the rank's weights per group and tick)"
                       f"{' (single-stream latency curve)' if pods == 1 else ''}, {'RCCL send/recv of the residual rows' if transport['kind'] == 'rccl' else 'residual rows staged through the host'} issued below the C-ABI (lh_pipeline_run)"
                       f"{'; SHARED-GPU functional mode (host-staged p2p)' if shared_gpu else ''}")
        result["transport"] = transport["kind"]
        model.free()

    Tbar = timed_pos0 + (K + 1) / 2.0
    wbytes, kvbytes = bytes_per_token(d, L, F, V, Tbar)
    if args.int8:  # matrices at 36/32 B per weight, norms + one embedding row stay f32
        mat = 4 * (L * (4 * d * d + 3 * d * F) + V * d)
        wbytes = (wbytes - mat) + mat * 36 // 128
    tok_s = tokens_total / dt
    rows_per_pass = max(1, pods // groups)   # streams that share one pass over a rank's weights (1: batch-1 decode)
    line = {
        "metric": "decode tokens/s LLaMA-7B fp32; % HBM-read roofline" if not args.int8 else "decode tokens/s LLaMA-7B block-int8 weights (config 4); % HBM-read roofline",
        "value": round(tok_s, 2), "unit": "tokens/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if not args.int8 else "f32 activations/accumulation, int8 block-quantised weights", "data": "synthetic (random-init weights, counter-based generator seed 1234; fixed 8-token prompt)",
        "config": {"workload": f"LLaMA-{args.shape} fp32 greedy decode, context {ctx_size}, positions {timed_pos0}..{timed_pos0 + K - 1}, batch 1 per stream",
                   "layers": L, "embd": d, "ff": F, "vocab": V, "streams": pods, "groups": groups, "parallelism": parallelism},
        "roofline_token": {
            "bytes_per_token": int(wbytes + kvbytes), "weights_bytes": int(wbytes),
            "roofline_tok_s_per_gpu_stream": round(HBM_PEAK_GBPS * 1e9 / (wbytes + kvbytes), 2),
            # a pass streams the weights once for all its rows and each row's own KV: bytes per pass / rows = bytes per token
            "rows_per_weight_pass": rows_per_pass,
            "frac_of_hbm_roofline": round(tok_s * (wbytes / rows_per_pass + kvbytes) / (HBM_PEAK_GBPS * 1e9 * world), 4),
        },
        "tokens_stream0": produced[: min(K, 16)],
    }
    if world == 1:
        line["value_definition"] = ("SURVEY 8d config 2: K single-token llama.Eval calls through the drop-in boundary; value = K / the sum of the HIP-event intervals around each "
                                    "lh_graph_compute (stream events at the call's start and behind its last launch); ms_per_step is that interval per call.  The wall clock of the "
                                    "same K steps between the barrier + synchronize pairs is eval_per_token_loop; the device-resident loop is resident_loop")
        line["timed_region_wall_s"] = round(dt_wall, 6)
    line.update(result)
    if rank == 0:
        print(json.dumps(line), file=_OUT, flush=True)
    if world > 1:
        dist.barrier()   # every rank is through its last exchange before any of them tears the control group down
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as e:   # whatever a rank dies of, it says so on one JSON line and exits non-zero: nobody waits for it
        import traceback
        traceback.print_exc()
        print(json.dumps({"error": f"rank {os.environ.get('RANK', '0')}: {e!r}", "n_gpus": int(os.environ.get("WORLD_SIZE", "1"))}), file=_OUT or sys.stdout, flush=True)
        sys.stdout.flush()
        os._exit(1)
