"""Stage-C smoke for kernel work: map-only pass, commit pass and a 2-shard split on small inputs, one line per step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from metalign_amd import synth
from metalign_amd._hip import Hip
hip = Hip.get(0)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
R, G = 100_000, 50
rng = np.random.default_rng(1)
src = rng.integers(1, G + 1, size=R)
recs = synth.make_alignment_records(src, G + 1, seed=3)
T = G + 1
d_recs, d_r2t = hip.array(recs), hip.array(np.arange(T, dtype=np.uint32))
d_acc = hip.empty(3 * T + 2, np.uint64)
if which in ("all", "map"):
    t0 = time.time()
    sh = hip.profile_begin_dev(d_recs.ptr, len(recs), False, d_r2t.ptr, T, T, 0.5)
    print("map-only:", sh.state_map(), sh.ngroups, "%.4f s" % (time.time() - t0), flush=True)
    sh.free()
if which in ("all", "commit"):
    t0 = time.time()
    sh = hip.profile_begin_dev(d_recs.ptr, len(recs), False, d_r2t.ptr, T, T, 0.5)
    sh.commit(True, True, 0, d_acc.ptr, d_acc.ptr + 8 * T, d_acc.ptr + 16 * T, d_acc.ptr + 24 * T, reset=True)
    hip.sync()
    print("commit ok %.4f s" % (time.time() - t0), flush=True)
    sh.free()
if which in ("all", "shard"):
    import oracle
    from test_gpu_fullsize import _stage_c_sharded
    oracle.build()
    starts = np.nonzero(recs["ref_new"] >> 31)[0]
    want = oracle.profile_assign(recs, np.arange(T, dtype=np.uint32), T, 0.5)
    got = _stage_c_sharded(hip, recs, np.arange(T, dtype=np.uint32), T, [int(starts[len(starts) // 2])])
    print("sharded:", all(np.array_equal(np.asarray(got[k]), np.asarray(want[k])) for k in want), flush=True)
