"""What a sample pays to get a sketch table onto the device: genome-major upload (radix sort + inversion on the device,
round 1) against the hash-major pairs as table format 2 stores them; filter built from the hashes against the stored bits.
python tools/table_upload_probe.py [ngenomes] [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import formats
from metalign_amd._hip import Hip
G = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
hip = Hip.get(0)
rng = np.random.default_rng(1)
h = np.sort(rng.integers(0, 1 << 62, size=(G, n), dtype=np.uint64), axis=1).reshape(-1)
o = np.arange(G + 1, dtype=np.uint64) * np.uint64(n)
ph, pg, gs = formats.pairs_from_genome_major(h, o)
def t(fn, reps=5):
    fn(); hip.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        x = fn(); x.free()
    hip.sync()
    return (time.perf_counter() - t0) / reps * 1e3
f = hip.filter_build(h); bits = f.download(); f.free()
print("%d genomes x %d hashes (%.0f MB of hashes):" % (G, n, h.nbytes / 1e6))
print("  upload_table (genome-major: radix sort on the device)  %.2f ms" % t(lambda: hip.upload_table(h, o)))
print("  upload_table_sorted (hash-major, as stored)            %.2f ms" % t(lambda: hip.upload_table_sorted(ph, pg, gs, int(ph[-1]))))
print("  filter_build (from the hashes)                         %.2f ms" % t(lambda: hip.filter_build(h)))
print("  filter_from_bits (stored, %.0f MB)                       %.2f ms" % (bits.nbytes / 1e6, t(lambda: hip.filter_from_bits(bits))))
