import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from metalign_amd import synth
from metalign_amd._hip import Hip
R, G, L = 12_500_000, 200_000, 5_000
ks = [21, 31, 51]
hip = Hip.get(0)
gb, go = synth.make_genomes(G, L)
rb, ro, _ = synth.make_reads(gb, go, R, npresent=max(50, G // 20))
d_b, d_o = hip.array(rb), hip.array(ro)
tables = [hip.sketch_genomes(gb, go, k, 1000)[0] for k in ks]
filts = [hip.filter_build(t) for t in tables]
hmaxs = [int(t.max()) for t in tables]
hip.stage_a_side_stream(True)
for per_cu in (3, 2, 1):
    hip.stage_a_workgroups_per_cu(per_cu)
    for hm, label in ((hmaxs, "dense"), ([int(3e-5 * 2 ** 64)] * 3, "hash only")):
        for rep in range(3):
            if rep == 1:
                hip.sync(); hip.prof_reset(); hip.prof_enable(True)
            hip.stage_a_side_stream(True)
            sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, R, ks, hm, 0, filts if label == "dense" else None)
            for sk in sks: sk.size
            for sk in sks: sk.free()
        hip.sync(); hip.prof_enable(False)
        c, t = hip.prof_get("sketch_reads")
        print("%d workgroups per CU, %s: %.2f ms" % (per_cu, label, t / 2), flush=True)
