"""How the two exchange schedules (one pass at a time / four passes in flight) react to collective latency that one
GPU cannot show: every collective of the path is preceded by a spin kernel of `delay` microseconds on the stream it
synchronises with (torch.cuda._sleep), world size 1 under RCCL.
Usage: python tools/exchange_latency_probe.py            (prints ms per pass for delays 0 / 100 / 300 us)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
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

# cycles per microsecond of the spin kernel's clock
torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize()
CYC_PER_US = 20_000_000 / ((time.perf_counter() - t0) * 1e6)
delay_us = [0]


def delayed(fn):
    def wrap(*a, **k):
        if delay_us[0]:
            torch.cuda._sleep(int(delay_us[0] * CYC_PER_US))
        return fn(*a, **k)
    return wrap


for name in ("all_gather", "all_gather_into_tensor", "all_to_all_single", "all_reduce"):
    setattr(dist, name, delayed(getattr(dist, name)))

for mode in ("0", "1"):
    os.environ["MG_EXCHANGE_PIPELINE"] = mode
    for d in (0, 100, 300):
        delay_us[0] = d
        job.run(6)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        job.run(30)
        torch.cuda.synchronize()
        print("%s  +%3d us per collective: %.3f ms per pass" % ("four passes in flight" if mode == "1" else "one pass at a time  ", d,
                                                                 (time.perf_counter() - t0) / 30 * 1e3), flush=True)
dist.destroy_process_group()
