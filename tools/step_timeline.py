"""Host-side timeline of one bench step (where the non-kernel time goes).  Usage: python tools/step_timeline.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from metalign_amd import distributed as mgd  # noqa: E402
from metalign_amd._hip import Hip  # noqa: E402

args = bench.parse()
hip = Hip.get(0)
w = bench.build_workload(args, 0, hip)
job = mgd.ShardJob(hip, None, 0, 1, k=args.k)
job.load(w["rb"], w["ro"], w["recs"], w["ref2tax"], w["dbh"], w["dbo"])
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
for _ in range(N):
    t = time.perf_counter()
    eng.profile_begin(0.5, False); eng.profile_commit_launch(1, True, 0); t = tic("stage C queued", t)
    sk = eng.sketch_local_async(job.k, job.hmax, 0); t = tic("stage A queued (no sync)", t)
    res = eng.containment_and_commit_results(sk, 2, False); t = tic("stage B queued + the step's one sync + read-backs", t)
    n = sk.size; sk.free(); t = tic("sketch size/free", t)
hip.sync()
tot = (time.perf_counter() - t_all) / N
for k, v in acc.items():
    print("%-40s %8.1f us" % (k, 1e6 * v / N))
print("%-40s %8.1f us" % ("step total", 1e6 * tot))
