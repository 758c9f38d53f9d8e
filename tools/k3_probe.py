"""Stage-C timing probe: R reads (1.25 records each) whose taxa are drawn from `present` of G genomes.
Usage: python tools/k3_probe.py R G present"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import synth
from metalign_amd._hip import Hip
R, G, present = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
hip = Hip.get(0)
rng = np.random.default_rng(1)
pres = rng.choice(np.arange(1, G + 1), size=present, replace=False)
src = pres[rng.integers(0, present, size=R)]
recs = synth.make_alignment_records(src, G + 1, seed=3)
T = G + 1
d_recs, d_r2t = hip.array(recs), hip.array(np.arange(T, dtype=np.uint32))
d_acc = hip.empty(3 * T + 2, np.uint64)
def run(commit=True):
    d_acc.memset(0)
    sh = hip.profile_begin_dev(d_recs.ptr, len(recs), False, d_r2t.ptr, T, T, 0.5)
    if commit:
        sh.commit(True, True, 0, d_acc.ptr, d_acc.ptr + 8 * T, d_acc.ptr + 16 * T, d_acc.ptr + 24 * T)
    else:
        sh.state_map()
    sh.free()
for commit in (True, False):
    run(commit); hip.sync()
    hip.prof_reset(); hip.prof_enable(True)
    for _ in range(5): run(commit)
    hip.sync(); hip.prof_enable(False)
    for nm in ("profile_pass", "profile_map"):
        n, t = hip.prof_get(nm)
        if n: print("R=%d G=%d present=%d %s: %.4f ms  (%.0f GB/s)" % (R, G, present, nm, t / n, len(recs) * 16 / (t / n) / 1e6))
