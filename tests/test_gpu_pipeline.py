"""-m gpu: the N > 1 layer-shard pipeline, end to end on ONE GPU, scheduled below the C-ABI (lh_pipeline_run).
 - two / three ranks share device 0 (p2p through the library's host-staged hook transport over gloo — RCCL cannot place two ranks
   on one device), each owning a block of layers, the residual stream and the token ids hopping between them.  Every stream must
   reproduce exactly the token ids of the single-process run, with the pipeline full (pods >= ranks) and as a single stream.
 - the RCCL transport itself (dlopen, ncclCommInitRank, grouped ncclSend/ncclRecv on the context's stream) runs with a world of
   one rank: the token id travels last stage -> first stage through RCCL on the one GPU a test box has."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def last_json(text):
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return json.loads(line)
    raise AssertionError("no JSON line in output:\n" + text[-2000:])


def only_json(text):
    """bench.py's contract: rank 0 prints ONE JSON line on stdout and nothing else (native libraries that print there - gloo's
    "[Gloo] Rank 0 is connected ..." - are kept off it)."""
    lines = [ln for ln in text.splitlines() if ln.strip()]
    assert len(lines) == 1, "stdout must be exactly one line:\n" + text[-2000:]
    return json.loads(lines[0])


def test_two_rank_pipeline_reproduces_single_process_tokens(product):
    args = ["--shape", "small", "--steps", "6", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = only_json(r1.stdout)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env2 = dict(env, BENCH_SHARED_GPU="1")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args,
                        cwd=ROOT, env=env2, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    two = only_json(r2.stdout)
    assert two["n_gpus"] == 2 and two["config"]["streams"] == 2   # default since round 6: one stream per rank, one row per weight pass (comparable with N = 1)
    assert two["roofline_token"]["rows_per_weight_pass"] == 1 and two["scaling_reference"]["ideal"] == "2 x value(N = 1)"
    assert two["tokens_stream0"] == one["tokens_stream0"], (one["tokens_stream0"], two["tokens_stream0"])
    assert len(one["tokens_stream0"]) == 6
    # the side objects: 4 N streams in 4-row passes (the headline of rounds 3-5), one stream walking through the stages
    assert two["pods_batched_4n"]["streams"] == 8 and two["pods_batched_4n"]["rows_per_weight_pass"] == 4 and two["pods_batched_4n"]["all_streams_equal_value_stream0"] is True
    assert two["single_stream"]["ids_match_batched_stream0"] is True


def _bench(args, env, nranks, launcher):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.join(ROOT, "bench.py"), "--gpus", str(nranks)] + args
    else:  # bench.py spawns its own ranks
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(nranks)] + args
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return only_json(r.stdout)


def test_single_stream_and_self_spawned_ranks(product):
    """`bench.py --gpus 3 --pods 1` without a launcher: one greedy stream walking through three stages (SURVEY §8e single-stream
    curve), ids equal to the 1-GPU run; and block-int8 weights through the same pipeline."""
    args = ["--shape", "small", "--steps", "5", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    one = _bench(args, env, 1, "self")
    env2 = dict(env, BENCH_SHARED_GPU="1")
    three = _bench(args + ["--pods", "1"], env2, 3, "self")
    assert three["n_gpus"] == 3 and three["config"]["streams"] == 1
    assert three["tokens_stream0"][:5] == one["tokens_stream0"][:5]
    q1 = _bench(args + ["--int8"], env, 1, "self")
    q2 = _bench(args + ["--int8", "--pods", "2"], env2, 2, "self")
    assert q2["tokens_stream0"][:5] == q1["tokens_stream0"][:5]


@pytest.mark.parametrize("sample", [0, 1])
def test_pipeline_two_ranks_share_one_gpu_greedy_and_sampled(product, sample):
    """Two ranks (two layers each) on ONE GPU, five streams as two groups: greedy ids == llama_GreedyDecode of every prompt alone, and with
    lh_pipeline_run_sample (the sampler on the LAST rank, its ring seeded from the prompts given there) == llama_SampleDecode alone."""
    import numpy as np
    from llama_go_amd.mlapi import make_hparams, SHAPES
    rng = np.random.default_rng(8)
    prompts = [[int(t) for t in rng.integers(0, SHAPES["small"]["vocab"], n)] for n in (5, 1, 9, 3, 2)]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "pipeline_worker.py"), "small", "40", "3", str(sample), json.dumps(prompts)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-8000:])
    got = last_json(r.stdout)
    assert got["groups"] == 2
    hp = make_hparams(**SHAPES["small"], ctx=40)
    m = product.NewSyntheticModel(hp, 17)
    for i, pr in enumerate(prompts):
        c = m.NewContext(40, 1)
        want = c.SampleDecode(pr, 6, topK=40, topP=0.95, temp=0.8, repeatPenalty=1.10, seed=777) if sample else c.GreedyDecode(pr, 6, want_logits=False)[0]
        c.free()
        assert got["ids"][i] == list(want), (i, got["ids"][i], want)
    m.free()


def test_a_failing_rank_takes_its_peers_down_instead_of_leaving_them_waiting(product):
    """Rank 0 is handed a token id outside the vocabulary: it refuses the run and aborts the communicator (lh_comm_abort).  On the
    host-staged transport that is the hooks' abort callback (a poison message); rank 1, which is waiting for rank 0's rows, must fail its
    run within seconds - with RCCL it is ncclCommAbort that does this."""
    from llama_go_amd.mlapi import SHAPES
    prompts = [[1, 2, 3], [4, SHAPES["small"]["vocab"] + 5]]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "pipeline_worker.py"), "small", "40", "3", "2", json.dumps(prompts)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)   # (a hang would hit the 120 s gloo timeout at best)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.strip().startswith("{") and '"rank"' in l]
    by_rank = {d["rank"]: d for d in lines}
    assert set(by_rank) == {0, 1}, (r.stdout[-1500:], r.stderr[-3000:])
    assert by_rank[0]["failed"] and "vocabulary" in by_rank[0]["error"]
    assert by_rank[1]["failed"], by_rank[1]


def test_rccl_transport_world_of_one(product, oracle):
    """lh_comm_unique_id / lh_comm_init / lh_comm_exchange on real RCCL with the one GPU of the box: a world of one rank sends the
    produced token id to itself (grouped ncclSend + ncclRecv on the context's stream).  Streams must equal the checker's greedy ids."""
    from llama_go_amd.mlapi import Pipeline, comm_unique_id, make_hparams, SHAPES
    hp = make_hparams(**SHAPES["tiny"], ctx=48)
    prompts = [[1, 5, 9, 200], [7, 3], [100, 101, 102, 103, 104, 105, 106, 107, 108, 109, 110]]
    m = product.NewSyntheticModel(hp, 77)
    cid = comm_unique_id(product)
    assert len(cid) == 128 and any(cid)
    pl = Pipeline(m, 48, len(prompts), 0, 1, comm_id=cid)
    pl.run(prompts, 3)
    # the diagnosis calls of the N > 1 bench line on a real RCCL communicator: ring shifts (here: grouped send / receive to self) and the
    # per-tick stage / exchange events
    hop = pl.hop_probe(4096 * 4, 50)
    assert 0.0 < hop < 1e5, hop
    pl.profile(True)
    pl.run(None, 4)
    st = pl.stats()
    assert st["ticks"] >= 4 and st["stage_ms"] > 0.0 and st["exchange_ms"] >= 0.0, st
    pl.profile(False)
    got = [pl.tokens(i) for i in range(len(prompts))]
    pl.free()
    # the same streams without any communicator (direct device copy of the id)
    pl2 = Pipeline(m, 48, len(prompts), 0, 1)
    pl2.run(prompts, 7)
    got2 = [pl2.tokens(i) for i in range(len(prompts))]
    pl2.free()
    m.free()
    om = oracle.NewSyntheticModel(hp, 77)
    for i, pr in enumerate(prompts):
        oc = om.NewContext(48, 4, False)
        want, _ = oc.GreedyDecode(pr, 8, want_logits=False)
        oc.free()
        assert got[i] == want, (i, got[i], want)
        assert got2[i] == want
    om.free()


def test_pipeline_argument_errors(product):
    from llama_go_amd.mlapi import Pipeline, MLError, make_hparams, SHAPES
    hp = make_hparams(**SHAPES["tiny"], ctx=32)
    m = product.NewSyntheticModel(hp, 5)
    with pytest.raises(MLError):
        Pipeline(m, 32, 2, 0, 2)           # sharded world without a communicator
    pl = Pipeline(m, 32, 2, 0, 1)
    with pytest.raises(MLError):
        pl.run(None, 2)                    # nothing to continue from
    with pytest.raises(MLError):
        pl.run([[1, 2], [9999]], 1)        # token id outside the vocabulary
    pl.run([[1, 2], [3]], 40)              # past the window of 32: an unsharded pipeline swaps context like server.Do (tests/test_context_swap.py)
    assert len(pl.tokens(0)) == 41 and len(pl.tokens(1)) == 41
    pl.run([[1, 2], [3]], 2)
    assert len(pl.tokens(0)) == 3 and len(pl.tokens(1)) == 3
    pl.free()
    half = product.NewSyntheticModel(hp, 5, 0, 1)
    with pytest.raises(MLError):
        Pipeline(half, 32, 1, 0, 1)        # half of the layers is not rank 0 of a world of one
    half.free()
    m.free()


def test_real_rccl_two_ranks_when_the_box_has_two_gpus(product):
    """RCCL with MORE THAN ONE rank has never run in this project's history (every gpurun box has one GPU): the first box with two GPUs validates
    it inside `pytest -m gpu`.  `bench.py --gpus 2` on the full 7B model, one rank per GPU, ncclCommInitRank + grouped ncclSend / ncclRecv of the
    residual rows on the compute streams (csrc/comm.hip) - stream 0's ids must be the committed single-GPU ids (tests/golden/7b_seed1234_ids.json,
    checked against the oracle when they were written), every stream equal to it, and the single-stream walk through the stages too."""
    product.lib.llamago_DeviceCount.restype = __import__("ctypes").c_int
    if product.lib.llamago_DeviceCount() < 2:
        pytest.skip("one GPU visible: RCCL with two ranks needs two (the shared-GPU tests above cover the scheduler over the host-staged transport)")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_SHARED_GPU"):
        env.pop(k, None)
    two = _bench(["--steps", "8", "--warmup", "2", "--no-cpu-baseline"], env, 2, "torchrun")
    assert two["n_gpus"] == 2 and two["config"]["streams"] == 2
    assert "SHARED-GPU" not in two["config"]["parallelism"]
    assert two["parity"]["ids_match_single_gpu"] is True and two["parity"]["all_streams_equal"] is True, two["parity"]
    assert two["single_stream"]["ids_match_batched_stream0"] is True, two["single_stream"]
    assert two["pods_batched_4n"]["all_streams_equal_value_stream0"] is True, two["pods_batched_4n"]
