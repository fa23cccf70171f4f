"""Layer-sharded pipeline driver: one process per GPU, contiguous blocks of layers per rank, the fp32
residual stream [n x embd] hops rank r -> r+1 over RCCL point-to-point (xGMI), the sampled token id returns from the
last rank to rank 0.  No all-reduce: a pure layer shard has exactly one exchange per stage boundary.

The reference's only data-parallel dimension is request-level "pods" (pkg/server/server.go:88-101: up to MaxPods
concurrent Do() goroutines, each with its own llama.Context over the shared Model).  Pods are what fills a pipeline:
with P >= R independent greedy streams in flight every rank is busy every tick.

Schedule ("tick" = one stage evaluation per rank):
    rank r evaluates stream p = (t - r) mod P at its step s = (t - r) div P           for 0 <= t - r < P * S
after every tick all ranks exchange in ONE grouped p2p batch (ring shift): rank r sends what it just produced to r+1
(the last rank sends the token id to rank 0) and receives what r-1 produced in the same tick.  Grouping the send and
the receive removes the circular wait a ring of blocking sends would have (RCCL send may block until the matching
receive is posted).  With P == R the token of stream p / step s reaches rank 0 exactly one tick before it is needed.

This module is transport-agnostic: `PipelineRunner` takes the stage function and tensors; bench.py binds it to
lh_llama_stage + torch.distributed(nccl == RCCL); tests/test_pipeline_gloo.py runs the same code over gloo on CPU.
"""
from dataclasses import dataclass
from typing import Callable, List, Optional


@dataclass
class Tick:
    t: int
    active: bool          # this rank evaluates a stage in this tick
    stream: int = -1
    step: int = -1
    recv_after: bool = False   # the previous rank was active in this tick -> a message arrives after it
    recv_stream: int = -1
    recv_step: int = -1


def schedule(rank: int, world: int, pods: int, steps: int) -> List[Tick]:
    """Ticks of one phase (S steps for each of P streams) as seen by `rank`."""
    if pods < world:
        raise ValueError(f"pods ({pods}) must be >= ranks ({world}): a stream's next token is only known after {world} ticks")
    total = pods * steps + world - 1
    out = []
    prev = (rank - 1) % world
    for t in range(total):
        k = t - rank
        tk = Tick(t=t, active=0 <= k < pods * steps)
        if tk.active:
            tk.stream, tk.step = k % pods, k // pods
        kp = t - prev
        if 0 <= kp < pods * steps:
            tk.recv_after, tk.recv_stream, tk.recv_step = True, kp % pods, kp // pods
        out.append(tk)
    return out


class PipelineRunner:
    """Runs phases of the schedule.  Callbacks:
        stage(stream, step, phase)         evaluate this rank's layers for (stream, step); inputs/outputs live in the
                                           per-stream buffers the callbacks below hand out
        send_buf(stream, step, phase)      tensor this rank sends after evaluating (x_out, or the token id on the last rank)
        recv_buf(stream, step, phase)      tensor that receives what the previous rank produced for (stream, step)
        on_recv(stream, step, phase)       optional hook after a message has been posted (e.g. keep a copy)
    `dist` is torch.distributed (or None for world == 1)."""

    def __init__(self, rank: int, world: int, pods: int, dist, stage: Callable, send_buf: Callable, recv_buf: Callable,
                 on_recv: Optional[Callable] = None):
        self.rank, self.world, self.pods, self.dist = rank, world, pods, dist
        self.stage, self.send_buf, self.recv_buf, self.on_recv = stage, send_buf, recv_buf, on_recv
        self.next, self.prev = (rank + 1) % world, (rank - 1) % world

    def run_phase(self, steps: int, phase: str):
        d = self.dist
        for tk in schedule(self.rank, self.world, self.pods, steps):
            if tk.active:
                self.stage(tk.stream, tk.step, phase)
            if self.world == 1:
                if tk.active and self.on_recv:
                    self.on_recv(tk.stream, tk.step, phase)
                continue
            ops = []
            if tk.active:
                ops.append(d.P2POp(d.isend, self.send_buf(tk.stream, tk.step, phase), self.next))
            if tk.recv_after:
                ops.append(d.P2POp(d.irecv, self.recv_buf(tk.recv_stream, tk.recv_step, phase), self.prev))
            if ops:
                for req in d.batch_isend_irecv(ops):
                    req.wait()
            if tk.recv_after and self.on_recv:
                self.on_recv(tk.recv_stream, tk.recv_step, phase)


def layer_range(rank: int, world: int, layers: int):
    """Contiguous block of layers owned by `rank` (7B: 32/16/8/4 layers for 1/2/4/8 ranks)."""
    return rank * layers // world, (rank + 1) * layers // world


class HostStagedDist:
    """torch.distributed look-alike for FUNCTIONAL testing of the N > 1 path when every rank shares one GPU (RCCL refuses two
    ranks on one device): device tensors are bounced through host memory and exchanged over gloo.  Not a performance path."""

    class P2POp:
        def __init__(self, op, tensor, peer):
            self.op, self.tensor, self.peer = op, tensor, peer

    def __init__(self, dist):
        self.d = dist
        self.isend, self.irecv = "isend", "irecv"

    def batch_isend_irecv(self, ops):
        reqs, post = [], []
        for o in ops:
            if o.op == "isend":
                host = o.tensor.detach().cpu()
                post.append((None, host))  # keep the host copy alive until the send completed
                reqs.append(self.d.isend(host, o.peer))
            else:
                host = o.tensor.detach().cpu()
                reqs.append(self.d.irecv(host, o.peer))
                post.append((o.tensor, host))

        class _Done:
            def __init__(self, reqs, post):
                self.reqs, self.post, self.done = reqs, post, False

            def wait(self):
                if self.done:
                    return
                for r in self.reqs:
                    r.wait()
                for dev, host in self.post:
                    if dev is not None:
                        dev.copy_(host)
                self.done = True

        return [_Done(reqs, post)]
