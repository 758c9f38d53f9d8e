"""Where the multi-GPU exchange path spends host time (1 rank, collectives forced).
torchrun --nproc-per-node 1 tools/exchange_timeline.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29521")
import torch
import torch.distributed as dist

import bench  # noqa: E402
from metalign_amd import distributed as mgd  # noqa: E402
from metalign_amd._hip import Hip  # noqa: E402

torch.cuda.set_device(0)
_stream = torch.cuda.Stream()
torch.cuda.set_stream(_stream)
dist.init_process_group("nccl", rank=0, world_size=1)
sys.argv = sys.argv[:1]
cfg = dict(bench.PRESETS[1], config=1)  # BASELINE configs[1]
hip = Hip.get(0, stream=_stream.cuda_stream)
w = bench.build_workload(cfg, 1000, 0, hip)
job = bench.make_job(hip, dist, 0, 1, cfg, w, force_dist=True)
acc = {}


def timed(obj, name, label=None):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        acc[label or name] = acc.get(label or name, 0.0) + time.perf_counter() - t0
        return r
    setattr(obj, name, wrap)


for n in ("sketch_local", "profile_begin", "export_sketch", "split_sketch", "merge_sketches", "containment", "profile_commit"):
    timed(job.engine, n)
timed(job, "_all_to_all")
timed(dist, "all_gather", "dist.all_gather")
timed(dist, "all_reduce", "dist.all_reduce")
for _ in range(3):
    job.step()
acc.clear()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    job.step()
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / N
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("%-24s %8.1f us" % (k, 1e6 * v / N))
print("%-24s %8.1f us (with the extra synchronisation of this tool)" % ("step total", 1e6 * tot))
dist.destroy_process_group()
