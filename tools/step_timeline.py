"""Host-side timeline of one bench step (where the non-kernel time goes).  Usage: python tools/step_timeline.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from metalign_amd import distributed as mgd  # noqa: E402
from metalign_amd._hip import Hip  # noqa: E402

cfg = dict(bench.PRESETS[1], config=1)  # BASELINE configs[1]
hip = Hip.get(0)
w = bench.build_workload(cfg, 1000, 0, hip)
job = bench.make_job(hip, None, 0, 1, cfg, w)
eng = job.engine
for _ in range(3):
    job.step()
acc = {}


def tic(name, t0):
    hip_t = time.perf_counter()
    acc[name] = acc.get(name, 0.0) + (hip_t - t0)
    return hip_t


N = 20
hip.sync()
t_all = time.perf_counter()
for _ in range(N):  # one pass at a time (bench.py keeps two in flight): host time of the two halves of a pass
    t = time.perf_counter()
    q = eng.queue_pass(0, job.ks, job.hmaxs, job.s, job.ci, job.pct_id, False); t = tic("pass queued (stage A, B, C; no sync)", t)
    sks, (hits, sizes), committed = eng.finish_pass(q, False); t = tic("the pass's one sync + read-backs", t)
hip.sync()
tot = (time.perf_counter() - t_all) / N
for k, v in acc.items():
    print("%-40s %8.1f us" % (k, 1e6 * v / N))
print("%-40s %8.1f us" % ("step total", 1e6 * tot))
