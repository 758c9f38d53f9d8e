"""Soak: many passes of the pipelined single-shard job; device memory in use must not creep and the last pass must
equal the first.  Usage: python tools/soak.py [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from metalign_amd import distributed as mgd
from metalign_amd._hip import Hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sys.argv = sys.argv[:1]
args = bench.parse()
hip = Hip.get(0)
w = bench.build_workload(args, 0, hip)
job = mgd.ShardJob(hip, None, 0, 1, k=args.k)
job.load(w["rb"], w["ro"], w["recs"], w["ref2tax"], w["dbh"], w["dbo"])
first = job.run(10, want_multimapped=True)
hip.sync()
import subprocess
def used():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True).stdout
    try:
        return int(out.strip().splitlines()[1].split(",")[2]) / 2**20
    except Exception:
        return -1
m0 = used()
t0 = time.perf_counter()
last = job.run(n, want_multimapped=True)
hip.sync()
dt = time.perf_counter() - t0
m1 = used()
same = all(np.array_equal(first[k], last[k]) for k in ("hits", "sizes", "count", "bases", "first_seen")) and \
    all(np.array_equal(a, b) for a, b in zip(first["multimapped"], last["multimapped"]))
print("passes %d  %.4f ms/pass  VRAM used %.0f -> %.0f MiB  identical results: %s" % (n, dt / n * 1e3, m0, m1, same))
