"""Stage A alone on configs[2]'s reads: the fused launch with the real thresholds / filters, and with thresholds that let
nothing through (no candidate handling at all: what the kernel costs as pure hashing).
python tools/k1_probe.py [nreads] [ngenomes] [genome_len] [ks, comma separated]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import synth
from metalign_amd import _hip
from metalign_amd._hip import Hip
R = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
L = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
ks = [int(x) for x in sys.argv[4].split(',')] if len(sys.argv) > 4 else [21, 31, 51]
hip = Hip.get(0)
if os.environ.get("MG_PROBE_HASH_MODE"):  # 1: min(hash(kmer), hash(revcomp)) % 9999999999971 — two hashes per k-mer
    hip.set_hash_mode(int(os.environ["MG_PROBE_HASH_MODE"]))
if os.environ.get("MG_PROBE_CS"):  # counters saturate here (default 3): fewer memory-side atomics per distinct hash at 1
    hip.count_saturation(int(os.environ["MG_PROBE_CS"]))
gb, go = synth.make_genomes(G, L)
rb, ro, _ = synth.make_reads(gb, go, R, npresent=max(50, G // 20))
d_b, d_o = hip.array(rb), hip.array(ro)
tables = [hip.sketch_genomes(gb, go, k, 1000)[0] for k in ks]
if os.environ.get("MG_PROBE_PREFIX") == "1":  # hash definition 1's prefix tables for k < kmax: every read k-mer is a candidate
    tables = [hip.sketch_genomes_prefix(gb, go, ks[-1], k, 1000)[0] for k in ks[:-1]] + tables[-1:]
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
    tail = " + ".join("%s %.2f" % (nm, hip.prof_get(nm)[1] / 2) for nm in ("table_clear", "bucket_sort", "bucket_pack"))
    print("%-46s stage A %.3f ms per pass (%d launch(es) per pass; %s ms), sketch sizes %s" % (label, t / 2, c // 2, tail, n), flush=True)
run(hmaxs, filts, "thresholds + filters of the table:")
if os.environ.get("MG_PROBE_RESIDENT", "1") != "0":
    res = [hip.filter_build(t) for t in tables]
    ok = [f.make_resident(t, hm, int(os.environ.get("MG_PROBE_SPREAD", "0"))) for f, t, hm in zip(res, tables, hmaxs)]
    print("resident indexes: %s, %.2f GB" % (ok, sum(f.resident_bytes for f in res) / 1e9), flush=True)
    run(hmaxs, res, "thresholds + resident indexes of the table:")
    for ab, what in ((1, "a flush drops its candidates"), (2, "looks them up, counts nothing, none goes round again"),
                     (3, "looks them up, counts nothing")):
        _hip.debug_set("resident_ablate", ab)
        run(hmaxs, res, "resident, %s:" % what)
    _hip.debug_set("resident_ablate", 0)
    for f in res: f.free()
run(hmaxs, None, "thresholds, no filter:")
span = 9999999999971 if os.environ.get("MG_PROBE_HASH_MODE") == "1" else 2 ** 64
run([int(3e-5 * span)] * len(ks), None, "thresholds that pass ~45 k k-mers per k (pure hashing):")
