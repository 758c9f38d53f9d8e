"""Stage A alone on configs[2]'s reads: the fused launch with the real thresholds / filters, and with thresholds that let
nothing through (no candidate handling at all: what the kernel costs as pure hashing).
python tools/k1_probe.py [nreads] [ngenomes] [genome_len] [ks, comma separated]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import synth
from metalign_amd._hip import Hip
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
ks = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [21, 31, 51]
hip = Hip.get(0)
gb, go = synth.make_genomes(G, L)
rb, ro, _ = synth.make_reads(gb, go, R, npresent=max(50, G // 20))
d_b, d_o = hip.array(rb), hip.array(ro)
tables = [hip.sketch_genomes(gb, go, k, 1000)[0] for k in ks]
filts = [hip.filter_build(t) for t in tables]
hmaxs = [int(t.max()) for t in tables]
def run(hm, fl, label):
    for rep in range(3):
        if rep == 1:
            hip.sync(); hip.prof_reset(); hip.prof_enable(True)
        sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, R, ks, hm, 0, fl)
        n = [sk.size for sk in sks]
        for sk in sks: sk.free()
    hip.sync(); hip.prof_enable(False)
    c, t = hip.prof_get("sketch_reads")
    print("%-46s stage A %.3f ms per pass (%d launch(es) per pass), sketch sizes %s" % (label, t / 2, c // 2, n), flush=True)
run(hmaxs, filts, "thresholds + filters of the table:")
run(hmaxs, None, "thresholds, no filter:")
run([int(3e-5 * 2 ** 64)] * len(ks), None, "thresholds that pass ~45 k k-mers per k (pure hashing):")
