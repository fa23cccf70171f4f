"""not-gpu: the layer-shard pipeline scheduler of the C-ABI (csrc/comm.hip: lh_pipeline_schedule, lh_pipeline_run_hooks) over
world_size-2/3 gloo on CPU.  The stage arithmetic is a stand-in (the HIP stage needs a GPU); what is verified is the N > 1
machinery bench.py relies on, driven by the SAME C loop lh_pipeline_run executes: routing of the residual stream rank r -> r+1,
return of the token id to rank 0, per-stream state, no deadlock (also with fewer streams than ranks: the single-stream curve),
and that the tokens equal a single-process evaluation of the same recurrence."""
import os
import subprocess
import sys
import textwrap

import pytest

from llama_go_amd.pipeline import layer_range, schedule

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_schedule_is_consistent_across_ranks(built):
    for world, pods, steps in [(1, 1, 3), (2, 2, 4), (2, 3, 2), (4, 4, 3), (8, 8, 2), (8, 11, 2), (4, 1, 3), (8, 1, 4), (4, 2, 3), (8, 3, 2), (3, 5, 0)]:
        sched = [schedule(r, world, pods, steps) for r in range(world)]
        Q = max(pods, world)
        nticks = (Q * (steps - 1) + pods + world - 1) if steps else 0
        assert all(len(s) == nticks for s in sched)
        for t in range(nticks):
            for r in range(world):
                me, nxt = sched[r][t], sched[(r + 1) % world][t]
                assert me.t == t
                # what r sends after tick t is exactly what r+1 expects to receive after tick t
                assert me.active == nxt.recv_after
                if me.active:
                    assert (me.stream, me.step) == (nxt.recv_stream, nxt.recv_step)
        # every (stream, step) is evaluated exactly once per rank, in dependency order
        for r in range(world):
            seen = [(tk.stream, tk.step) for tk in sched[r] if tk.active]
            assert sorted(seen) == sorted((p, s) for p in range(pods) for s in range(steps))
        # rank r+1 evaluates (p, s) exactly one tick after rank r; rank 0 evaluates (p, s+1) strictly after the last rank produced (p, s)
        when = [{(tk.stream, tk.step): tk.t for tk in sched[r] if tk.active} for r in range(world)]
        for r in range(world - 1):
            for key, t in when[r].items():
                assert when[r + 1][key] == t + 1
        for (p, s_), t in when[0].items():
            if s_ > 0:
                assert when[world - 1][(p, s_ - 1)] < t
        # with at least as many streams as ranks nobody idles between its first and last active tick
        if pods >= world and steps:
            for r in range(world):
                act = [tk.t for tk in sched[r] if tk.active]
                assert act == list(range(act[0], act[-1] + 1))


def test_streams_are_dealt_into_groups_that_fill_the_ranks(built):
    """lh_pipeline_group_count: min(pods, world) groups (every rank busy every tick, one weight pass per group), more when a group
    would exceed the rows one pass takes; pods = 4 world -> world groups of 4; max_rows = 1 -> every stream on its own."""
    import ctypes as C
    import llama_go_amd as pkg
    lib = C.CDLL(pkg.LIBLLAMAHIP)
    f = lib.lh_pipeline_group_count
    f.restype, f.argtypes = C.c_uint32, [C.c_uint32] * 3
    for world in (1, 2, 4, 8):
        assert f(4 * world, world, 0) == world
        assert f(1, world, 0) == 1
        assert f(3, world, 1) == 3
    assert f(100, 1, 0) == 2 and f(100, 1, 48) == 3 and f(64, 1, 0) == 1
    assert f(6, 1, 4) == 2 and f(7, 8, 0) == 7
    assert f(0, 1, 0) == 0
    for pods, world, mr in ((32, 8, 0), (100, 1, 48), (6, 1, 4), (7, 3, 2)):
        G = f(pods, world, mr)
        sizes = [sum(1 for p in range(pods) if p * G // pods == g) for g in range(G)]
        assert sum(sizes) == pods and min(sizes) >= 1 and max(sizes) <= (mr or 64) and max(sizes) - min(sizes) <= 1


def test_layer_ranges_partition_the_model():
    for L in (32, 40, 80):
        for world in (1, 2, 4, 8):
            rs = [layer_range(r, world, L) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == L
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
    assert [layer_range(r, 8, 32) for r in range(8)][3] == (12, 16)


WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import torch, torch.distributed as dist
    from llama_go_amd.pipeline import run_hooks
    rank, world, pods, steps = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), {pods}, {steps}
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    D = 8
    first, last = rank == 0, rank == world - 1
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    xin = [torch.zeros(D) for _ in range(pods)]; xout = [torch.zeros(D) for _ in range(pods)]
    tok_in = [torch.zeros(1, dtype=torch.int32) for _ in range(pods)]; tok_out = [torch.zeros(1, dtype=torch.int32) for _ in range(pods)]
    produced = [[] for _ in range(pods)]
    def stage(p, s):
        if first:
            tok = (p + 1) if s == 0 else int(tok_in[p][0])
            x = torch.arange(D, dtype=torch.float32) * 0.5 + tok + 0.25 * s
        else:
            x = xin[p].clone()
        x = x * 1.5 + (rank + 1)          # this rank's "layers"
        if last:
            tok_out[p][0] = int(x.sum().item()) % 1000
            produced[p].append(int(tok_out[p][0]))
        else:
            xout[p].copy_(x)
    def exchange(s, u, rs, ru):           # one grouped p2p per tick: post both sides, then wait
        reqs = []
        if s >= 0:
            reqs.append(dist.isend((tok_out[s] if last else xout[s]).clone(), nxt))
        if rs >= 0:
            reqs.append(dist.irecv(tok_in[rs] if first else xin[rs], prv))
        for r in reqs:
            r.wait()
    run_hooks(rank, world, pods, steps, stage, exchange)   # the C scheduler loop (lh_pipeline_run_hooks)
    dist.barrier()
    if last:
        print("RESULT " + json.dumps(produced), flush=True)
    dist.destroy_process_group()
""")


def reference_tokens(world, pods, steps):
    import torch
    out = []
    for p in range(pods):
        toks, tok = [], p + 1
        for s in range(steps):
            x = torch.arange(8, dtype=torch.float32) * 0.5 + tok + 0.25 * s
            for r in range(world):
                x = x * 1.5 + (r + 1)
            tok = int(x.sum().item()) % 1000
            toks.append(tok)
        out.append(toks)
    return out


@pytest.mark.parametrize("world,pods,steps", [(2, 2, 4), (3, 4, 3), (3, 1, 4), (8, 8, 3), (8, 1, 2)])   # world 8: the driver's largest run - eight groups of one, and one stream through eight stages
def test_pipeline_over_gloo_matches_single_process(built, tmp_path, world, pods, steps):
    import json
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, pods=pods, steps=steps, port=port))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("pipeline deadlocked")
        outs.append(o)
        assert p.returncode == 0, o
    line = [l for l in outs[-1].splitlines() if l.startswith("RESULT ")][0]
    assert json.loads(line[7:]) == reference_tokens(world, pods, steps)
