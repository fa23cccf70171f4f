"""Layer-sharded pipeline — thin Python side.

The schedule and its driver live BELOW the C-ABI (csrc/comm.hip: lh_pipeline_schedule / lh_pipeline_run_hooks /
lh_pipeline_run, RCCL send/recv in lh_comm_exchange), so that the Go host of the reference can run a layer-sharded model
through the shim (pkg/server/server.go:84-106 Engine + :151 one llama.Context per pod).  What is left here:

  schedule(...)        ctypes view of lh_pipeline_schedule (the pure function; no GPU needed)
  run_hooks(...)       lh_pipeline_run_hooks with Python stage/exchange callbacks: the C scheduler loop over any transport
                       (tests/test_pipeline_gloo.py drives it over gloo on CPU)
  layer_range(...)     contiguous block of layers per rank
  gloo_comm_hooks(...) an lh_comm_hooks transport over torch.distributed(gloo) host buffers, for FUNCTIONAL runs of the
                       real stages with several ranks on one GPU (RCCL refuses two ranks on one device); never a measurement.

Schedule: with Q = max(pods, world), rank r evaluates stream p at unit u in tick t = u*Q + p + r; after every tick all ranks
exchange once in ONE grouped p2p (ring shift): rank r sends what it just produced to r+1 (the last rank sends the token id to
rank 0) and receives what r-1 produced in the same tick.  pods >= world keeps every rank busy every tick (the reference's
request-level parallelism fills the pipeline); pods = 1 is the single greedy stream walking through the stages.
"""
import ctypes as C
from dataclasses import dataclass
from typing import Callable, List

from . import LIBLLAMAHIP


class _Tick(C.Structure):
    """lh_tick (include/llamahip.h)."""
    _fields_ = [("t", C.c_uint32), ("stream", C.c_int32), ("unit", C.c_int32), ("recv_stream", C.c_int32), ("recv_unit", C.c_int32)]


@dataclass
class Tick:
    t: int
    active: bool               # this rank evaluates a stage in this tick
    stream: int = -1
    step: int = -1
    recv_after: bool = False   # the previous rank was active in this tick -> a message arrives after it
    recv_stream: int = -1
    recv_step: int = -1


STAGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint32)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32)


class _PipelineHooks(C.Structure):
    """lh_pipeline_hooks."""
    _fields_ = [("user", C.c_void_p), ("stage", STAGE_FN), ("exchange", EXCHANGE_FN)]


COMM_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_uint64, C.c_int)


COMM_ABORT_FN = C.CFUNCTYPE(None, C.c_void_p)


class CommHooks(C.Structure):
    """lh_comm_hooks."""
    _fields_ = [("user", C.c_void_p), ("exchange", COMM_EXCHANGE_FN), ("abort", COMM_ABORT_FN)]


_lib = None


def _hip():
    """libllamahip.so (loads without a GPU; the schedule functions touch no device)."""
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIBLLAMAHIP, mode=C.RTLD_GLOBAL)
        _lib.lh_pipeline_schedule.restype = C.c_int
        _lib.lh_pipeline_schedule.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_Tick), C.c_uint32]
        _lib.lh_pipeline_run_hooks.restype = C.c_int
        _lib.lh_pipeline_run_hooks.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_PipelineHooks)]
    return _lib


def schedule(rank: int, world: int, pods: int, steps: int) -> List[Tick]:
    """Ticks of one run (`steps` units for each of `pods` streams) as seen by `rank` — lh_pipeline_schedule."""
    lib = _hip()
    n = lib.lh_pipeline_schedule(rank, world, pods, steps, None, 0)
    if n < 0:
        raise ValueError(f"lh_pipeline_schedule({rank}, {world}, {pods}, {steps}) = {n}")
    arr = (_Tick * max(n, 1))()
    lib.lh_pipeline_schedule(rank, world, pods, steps, arr, n)
    return [Tick(t=k.t, active=k.stream >= 0, stream=k.stream, step=k.unit, recv_after=k.recv_stream >= 0, recv_stream=k.recv_stream, recv_step=k.recv_unit)
            for k in arr[:n]]


def run_hooks(rank: int, world: int, pods: int, steps: int, stage: Callable[[int, int], None],
              exchange: Callable[[int, int, int, int], None]):
    """The C scheduler loop (lh_pipeline_run_hooks) with Python actions: stage(stream, unit) and
    exchange(send_stream, send_unit, recv_stream, recv_unit) with -1 for an absent side."""
    errs = []

    def _stage(_u, s, u):
        try:
            stage(s, u)
            return 0
        except BaseException as e:  # an exception must not unwind through the C frames
            errs.append(e)
            return -1

    def _exchange(_u, s, u, rs, ru):
        try:
            exchange(s, u, rs, ru)
            return 0
        except BaseException as e:
            errs.append(e)
            return -1

    hooks = _PipelineHooks(None, STAGE_FN(_stage), EXCHANGE_FN(_exchange))
    rc = _hip().lh_pipeline_run_hooks(rank, world, pods, steps, C.byref(hooks))
    if errs:
        raise errs[0]
    if rc:
        raise RuntimeError(f"lh_pipeline_run_hooks returned {rc}")


def layer_range(rank: int, world: int, layers: int):
    """Contiguous block of layers owned by `rank` (7B: 32/16/8/4 layers for 1/2/4/8 ranks)."""
    return rank * layers // world, (rank + 1) * layers // world


def gloo_comm_hooks(dist):
    """lh_comm_hooks over torch.distributed (gloo): the library stages each message through pinned host memory and calls
    exchange() once per tick with both directions; isend + irecv are posted together, then both are waited for."""
    import torch

    def _exchange(_user, send_p, send_n, send_peer, recv_p, recv_n, recv_peer):
        try:
            reqs, keep = [], []
            if send_p and send_n:
                src = (C.c_uint8 * send_n).from_address(send_p)
                t = torch.frombuffer(src, dtype=torch.uint8).clone()
                keep.append(t)
                reqs.append(dist.isend(t, send_peer))
            rt = None
            if recv_p and recv_n:
                rt = torch.empty(recv_n, dtype=torch.uint8)
                reqs.append(dist.irecv(rt, recv_peer))
            for r in reqs:
                r.wait()
            if rt is not None:
                C.memmove(recv_p, rt.data_ptr(), recv_n)
            return 0
        except BaseException as e:
            import sys
            print(f"gloo transport failed: {e!r}", file=sys.stderr, flush=True)
            return -1

    def _abort(_user):
        # lh_comm_abort: the successor is (or will be) waiting for this rank's rows - a one-byte message where it expects more makes its
        # receive fail (gloo: message size mismatch) instead of block; its own abort then passes the failure on round the ring
        try:
            world, rank = dist.get_world_size(), dist.get_rank()
            if world > 1:
                dist.isend(torch.zeros(1, dtype=torch.uint8), (rank + 1) % world)
        except BaseException:
            pass

    hooks = CommHooks(None, COMM_EXCHANGE_FN(_exchange), COMM_ABORT_FN(_abort))
    return hooks
