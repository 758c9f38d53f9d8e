"""Stage C on bench.py's configs[2] records (log-normal abundances over 500 present genomes, 10 001 taxa): time per pass and,
with the instrumented library, clocks per phase.   [MG_LIB_PATH=metalign_amd/libmetalign_hip_phases.so] python tools/k3_bench_shape.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import synth
from metalign_amd._hip import Hip
hip = Hip.get(0)
G = 10000
gb, go = synth.make_genomes(G, 1000)  # (short genomes: only the read -> genome assignment matters here)
rb, ro, src = synth.make_reads(gb, go, 10_000_000, npresent=500, seed=synth.SEED + 1)
recs = synth.make_alignment_records(src + 1, G + 1, seed=synth.SEED + 2)
names = ["ticket", "load+desc", "walk", "blockscan", "lookback", "publish", "commit", "looptop/exit"]
for T, r2t in ((G + 1, np.arange(G + 1, dtype=np.uint32)), (2001, (np.arange(G + 1) % 2001).astype(np.uint32))):
    d_recs, d_r2t = hip.array(recs), hip.array(r2t)
    d_acc = hip.empty(3 * T + 2, np.uint64)
    def run():
        d_acc.memset(0)
        sh = hip.profile_begin_dev(d_recs.ptr, len(recs), False, d_r2t.ptr, len(r2t), T, 0.5)
        sh.commit(True, True, 0, d_acc.ptr, d_acc.ptr + 8 * T, d_acc.ptr + 16 * T, d_acc.ptr + 24 * T)
        sh.free()
    run(); hip.sync()
    ph = hasattr(hip.lib, "mg_debug_k3_phases")
    out = (ctypes.c_ulonglong * 16)()
    if ph:
        hip.lib.mg_debug_k3_phases(out, 1)
    hip.prof_reset(); hip.prof_enable(True)
    for _ in range(5): run()
    hip.sync(); hip.prof_enable(False)
    n, t = hip.prof_get("profile_pass")
    print("records=%d T=%d profile_pass: %.4f ms (%.0f GB/s)" % (len(recs), T, t / n, len(recs) * 16 / (t / n) / 1e6))
    if ph:
        hip.lib.mg_debug_k3_phases(out, 0)
        nt = (len(recs) + 2047) // 2048 * 5
        print("   cycles/tile: " + ", ".join("%s %d" % (nm, out[8 + i] / nt) for i, nm in enumerate(names)))
    top = np.bincount(r2t[recs["ref_new"] & 0x7fffffff], minlength=T)
    print("   hottest taxa share of records:", np.round(np.sort(top)[::-1][:5] / len(recs), 3))
