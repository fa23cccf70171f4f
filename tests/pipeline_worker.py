"""Worker of tests/test_gpu_pipeline.py::test_sampled_pipeline_two_ranks_share_one_gpu: one rank of a layer-sharded pipeline whose
ranks all sit on device 0 (host-staged p2p over gloo, the library's hook transport).  Launched by torch.distributed.run; rank 0 prints
one JSON line with every stream's ids.  usage: pipeline_worker.py <shape> <ctx> <steps> <sample 0|1> <prompts as JSON>"""
import datetime
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (one HIP runtime per process: torch's)
import torch.distributed as dist  # noqa: E402

shape, ctx, steps, sample, prompts = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), json.loads(sys.argv[5])
keep = int(sys.argv[6]) if len(sys.argv) > 6 else None   # given: the streams outlive their windows (context swap with this KeepCount), second run of `more` steps
more = int(sys.argv[7]) if len(sys.argv) > 7 else 2
os.environ["LLAMAGO_DEVICE"] = "0"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
rank, world = dist.get_rank(), dist.get_world_size()
from llama_go_amd.mlapi import SHAPES, Pipeline, load_product, make_hparams  # noqa: E402
from llama_go_amd.pipeline import gloo_comm_hooks, layer_range  # noqa: E402

prod = load_product()
hp = make_hparams(**SHAPES[shape], ctx=ctx)
l0, l1 = layer_range(rank, world, hp.layersCount)
m = prod.NewSyntheticModel(hp, 17, l0, l1)
pl = Pipeline(m, ctx, len(prompts), rank, world, hooks=gloo_comm_hooks(dist))
smp = dict(topK=40, topP=0.95, temp=0.8, repeatPenalty=1.10, seed=777)
if keep is not None:
    pl.SetKeepCount(keep)
if sample == 2:
    # failure path: rank 0 sees a bad prompt and aborts the communicator; every other rank must FAIL its run (the transport's abort hook
    # poisons its pending receive) instead of waiting forever.  Every rank reports what happened to it on stderr-free stdout lines.
    try:
        pl.run(prompts, steps)
        print(json.dumps({"rank": rank, "failed": False}), flush=True)
    except Exception as e:
        print(json.dumps({"rank": rank, "failed": True, "error": str(e)[:200]}), flush=True)
    os._exit(0)
elif sample:
    pl.run_sample(prompts, steps, **smp)
    pl.run_sample(None, more, **smp)
else:
    pl.run(prompts, steps)
    pl.run(None, more)
if rank == 0:
    print(json.dumps({"ids": [pl.tokens(i) for i in range(len(prompts))], "groups": pl.groups}), flush=True)
pl.free()
m.free()
dist.barrier()   # no rank tears its gloo pairs down while the other is still inside the run
dist.destroy_process_group()
