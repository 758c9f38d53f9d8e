"""Soak of the exchange schedule (four passes in flight) with every collective in the path on one GPU (RCCL, world 1):
many passes, results must stay identical to the first pass and device memory must not creep.
Usage: python tools/soak_exchange.py [passes] [config]"""
import os, sys, time, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
import numpy as np
import torch, torch.distributed as dist
import bench
from metalign_amd import distributed as mgd
from metalign_amd._hip import Hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
torch.cuda.set_device(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
dist.init_process_group("nccl", rank=0, world_size=1)
config = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # BASELINE configs[1] by default
hip = Hip.get(0, stream=st.cuda_stream)
cfg = dict(bench.PRESETS[config], config=config)
cfg["match"] = "kmer" if mgd.kmer_match_by_default(max(cfg["ks"])) else "hash"  # (what bench.py runs by default: identity from k_max = 25 on)
w = bench.build_workload(cfg, 1000, 0, hip)
job = bench.make_job(hip, dist, 0, 1, cfg, w, force_dist=True)
first = job.run(10)
torch.cuda.synchronize()


def used():  # MiB of device memory in use, as the runtime reports it
    free, total, _ = hip.mem_info()
    return (total - free) / 2**20


m0 = used()
bad = 0
t0 = time.perf_counter()
for chunk in range(n // 250):
    last = job.run(250)
    if not all(np.array_equal(first[k], last[k]) for k in ("hits_k", "sizes_k", "count", "bases", "first_seen")):
        bad += 1
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("passes %d  %.4f ms/pass  VRAM used %.0f -> %.0f MiB  chunks with a differing result: %d" % (n // 250 * 250, dt / (n // 250 * 250) * 1e3, m0, used(), bad))
dist.destroy_process_group()
