"""Host (vectorised numpy over the downloaded CSR) vs device resolution of multimapped reads.
Usage: python tools/bench_multimap.py [R reads] [G genomes]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import map_and_profile as mp, synth  # noqa: E402
from metalign_amd._hip import Hip  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
hip = Hip.get(0)
rng = np.random.default_rng(1)
src = rng.integers(1, G + 1, size=R)
recs = synth.make_alignment_records(src, G + 1, seed=3)
T = G + 1
ref2tax = np.arange(T, dtype=np.uint32)
taxids = ["t%d" % i for i in range(T)]
taxid2info = {t: [50000.0] for t in taxids}
args = argparse.Namespace(verbose=False, length_normalize=False, low_mem=False)

t0 = time.perf_counter()
res = hip.profile_assign(recs, ref2tax, T, 0.5)
t1 = time.perf_counter()
a = {taxids[i]: [int(res["count"][i]), float(res["bases"][i])] for i in range(T) if res["count"][i] > 1}
mp.resolve_multi_prop_csr(args, a, dict(res, taxids=taxids), taxid2info)
t2 = time.perf_counter()
print("host  : assign + CSR download %.3f s, resolve %.3f s  (%d multimapped reads)" % (t1 - t0, t2 - t1, len(res["mm_hitlen"])))

t0 = time.perf_counter()
rd = hip.profile_assign_resident(recs, ref2tax, T, 0.5)
t1 = time.perf_counter()
b = {taxids[i]: [int(rd["count"][i]), float(rd["bases"][i])] for i in range(T) if rd["count"][i] > 1}
mp.resolve_multi_prop_device(args, b, dict(rd, taxids=taxids), taxid2info)
t2 = time.perf_counter()
rd["resident"].free()
print("device: assign (CSR stays in HBM) %.3f s, resolve %.3f s" % (t1 - t0, t2 - t1))
worst = max(abs(a[k][1] - b[k][1]) / max(1.0, abs(a[k][1])) for k in a)
print("largest relative difference of a taxon's bases: %.2e" % worst)
n, t = hip.prof_get("resolve_multimapped")
