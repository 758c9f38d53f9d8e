"""Where a stage-C tile spends its time: thread 0's shader clock between the phase boundaries of k_profile_pass,
averaged per tile, for the commit pass and the map-only pass.  Needs the instrumented build:
    make -C metalign_amd/csrc phases
    MG_LIB_PATH=metalign_amd/libmetalign_hip_phases.so python tools/k3_phases.py
(a second library next to the shipped one, which has no clocks in it)."""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import synth
from metalign_amd._hip import Hip
hip = Hip.get(0)
lib = hip.lib
if not hasattr(lib, "mg_debug_k3_phases"):
    sys.exit("build with K3_PHASES=1 first (see the docstring)")
names = ["ticket", "load+desc", "walk", "blockscan", "lookback", "publish", "flush+sync", "looptop/exit", "commit walk", "binning", "start", "last flush"]
for R, G, present in [(12500000, 2000, 2000), (10000000, 10000, 500)]:
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
        out = (ctypes.c_ulonglong * 24)()
        lib.mg_debug_k3_phases(out, 1)
        hip.prof_reset(); hip.prof_enable(True)
        for _ in range(5): run(commit)
        hip.sync(); hip.prof_enable(False)
        lib.mg_debug_k3_phases(out, 0)
        ntiles = (len(recs) + 2047) // 2048
        for nm in ("profile_pass", "profile_map"):
            n, t = hip.prof_get(nm)
            if n: print("R=%d T=%d %s: %.4f ms (%.0f GB/s) tiles=%d" % (R, T, nm, t / n, len(recs) * 16 / (t / n) / 1e6, ntiles))
        off = 12 if commit else 0
        tot = sum(out[off:off + 12])
        print("   cycles/tile: " + ", ".join("%s %.0f" % (names[i], out[off + i] / 5 / ntiles) for i in range(12)), " total %.0f" % (tot / 5 / ntiles))
    d_recs.free(); d_r2t.free(); d_acc.free()
