"""-m gpu: the N > 1 layer-shard pipeline of bench.py, end to end on ONE GPU: two ranks share device 0 (p2p bounced through the
host over gloo — RCCL cannot place two ranks on one device), each owning half of the layers, the residual stream and the sampled
token ids hopping between them through lh_llama_stage.  The streams must reproduce exactly the token ids of the single-process run."""
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


def test_two_rank_pipeline_reproduces_single_process_tokens(product):
    args = ["--shape", "small", "--steps", "6", "--warmup", "1", "--no-cpu-baseline"]
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = last_json(r1.stdout)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env2 = dict(env, BENCH_SHARED_GPU="1")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2"] + args,
                        cwd=ROOT, env=env2, capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    two = last_json(r2.stdout)
    assert two["n_gpus"] == 2 and two["config"]["streams"] == 8   # default: 4 streams per rank in flight
    assert two["tokens_stream0"] == one["tokens_stream0"], (one["tokens_stream0"], two["tokens_stream0"])
    assert len(one["tokens_stream0"]) == 6
