"""What ONE rank's merge step costs at 8 ranks on configs[3] shapes, emulated on one GPU: eight different read shards
(12.5M reads each by default) are sketched one after the other against the full 200k-genome thresholds / filters, the
slice of hash range 0 (1/8 of the table's hashes) is cut out of each, and the eight slices are merged exactly as the
exchange path does (mg_sketch_merge_dev_async).  Prints pairs in, union out, and the merge's kernel times.
python tools/merge_w8_probe.py [reads_per_rank] [genomes] [genome_len] [world]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import synth
from metalign_amd._hip import Hip
from metalign_amd.distributed import table_bounds, table_max_hash
R = int(sys.argv[1]) if len(sys.argv) > 1 else 12_500_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5_000
W = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ks = [21, 31, 51]
hip = Hip.get(0)
gb, go = synth.make_genomes(G, L)
tables = [hip.sketch_genomes(gb, go, k, 1000) for k in ks]
filts = [hip.filter_build(t[0]) for t in tables]
hmaxs = [table_max_hash(h, o) for h, o in tables]
bounds = [table_bounds(h, W, hm) for (h, _), hm in zip(tables, hmaxs)]
slices = [[] for _ in ks]
local = []
for r in range(W):
    t0 = time.time()
    rb, ro, _ = synth.make_reads(gb, go, R, npresent=max(50, G // 20), seed=synth.SEED + 1 + 1000 * r)
    d_b, d_o = hip.array(rb), hip.array(ro)
    for rep in range(2):  # the second call has the distinct-count hint of the first
        sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, R, ks, hmaxs, 0, filts)
        if rep == 0:
            for sk in sks: sk.free()
    sizes = []
    for ki, sk in enumerate(sks):
        h, c = sk.download()
        cut = sk.split([bounds[ki][1]])[0]
        slices[ki].append((h[:cut].copy(), c[:cut].copy()))
        sizes.append(len(h))
        sk.free()
    local.append(sizes)
    d_b.free(); d_o.free()
    print("rank %d: local sketch sizes %s, slice 0 sizes %s (%.0f s)" % (r, sizes, [len(slices[ki][-1][0]) for ki in range(len(ks))], time.time() - t0), flush=True)
for ki, k in enumerate(ks):
    rh = np.concatenate([s[0] for s in slices[ki]]); rc = np.concatenate([s[1] for s in slices[ki]])
    d_h, d_c = hip.array(rh), hip.array(rc)
    lo, hi = bounds[ki][0], bounds[ki][1]
    for rep in range(3):
        if rep == 1:
            hip.sync(); hip.prof_reset(); hip.prof_enable(True)
        m = hip.sketch_merge_dev_async(d_h.ptr, d_c.ptr, rh.size, k, lo, hi - 1)
        m.resolve(); n = m.size; m.free()
    hip.sync(); hip.prof_enable(False)
    t = {nm: hip.prof_get(nm) for nm in ("table_clear", "merge_insert", "bucket_sort", "bucket_pack", "merge_sort")}
    print("k=%d: %d pairs in (%d slices) -> union %d (%.2f of the sum); per merge: %s" %
          (k, rh.size, W, n, n / max(rh.size, 1), ", ".join("%s %.3f ms" % (nm, v[1] / 2) for nm, v in t.items() if v[0])), flush=True)
    d_h.free(); d_c.free()
