"""Soak: many passes of the pipelined single-shard job; device memory in use must not creep and the last pass must
equal the first.  Usage: python tools/soak.py [passes] [config]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from metalign_amd import distributed as mgd
from metalign_amd._hip import Hip
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
config = int(sys.argv[2]) if len(sys.argv) > 2 else 1  # BASELINE configs[1] by default (configs[2]: python tools/soak.py 300 2)
hip = Hip.get(0)
cfg = dict(bench.PRESETS[config], config=config)
cfg["match"] = "kmer" if mgd.kmer_match_by_default(max(cfg["ks"])) else "hash"  # (what bench.py runs by default: identity from k_max = 25 on)
w = bench.build_workload(cfg, 1000, 0, hip)
job = bench.make_job(hip, None, 0, 1, cfg, w)
MM = os.environ.get("SOAK_MM", "1") == "1"
first = job.run(10, want_multimapped=MM)
hip.sync()
def used():  # MiB of device memory in use, as the runtime reports it
    free, total, _ = hip.mem_info()
    return (total - free) / 2**20
m0 = used()
t0 = time.perf_counter()
trace = []
for part in range(6):  # VRAM in use along the way: a plateau is the pool filling up, a slope is a leak
    last = job.run(max(n // 6, 1), want_multimapped=MM)
    hip.sync()
    trace.append(round(used()))
dt = time.perf_counter() - t0
m1 = used()
same = all(np.array_equal(first[k], last[k]) for k in ("hits_k", "sizes_k", "count", "bases", "first_seen")) and \
    (not MM or all(np.array_equal(a, b) for a, b in zip(first["multimapped"], last["multimapped"])))
print("passes %d  %.4f ms/pass  VRAM used %.0f -> %.0f MiB (along the way: %s)  identical results: %s" % (n, dt / n * 1e3, m0, m1, trace, same))
