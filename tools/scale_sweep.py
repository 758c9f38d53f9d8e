"""Per-kernel time and achieved algorithmic GB/s of stages B and C at the sizes of BASELINE.json configs[1..3]
(per-GPU share).  Usage: python tools/scale_sweep.py [--out gpurun_out/scale.json] [--big]

Stage B (containment): table of G genomes x n hashes (random uniform below hmax, each genome ascending), read
sketch of Q distinct hashes of which half are table members.  Algorithmic bytes = G*n*8 (table) + Q*12 (sketch).
Stage C (assign + histogram): R reads, 1.25 records/read (synth.make_alignment_records), T = G+1 taxa.
Algorithmic bytes = 16 B per record.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import synth  # noqa: E402
from metalign_amd._hip import Hip, REC_DTYPE  # noqa: E402


def prof(hip, names):
    out = {}
    for nm in names:
        n, t = hip.prof_get(nm)
        if n:
            out[nm] = t / n
    return out


def stage_b(hip, G, n, Q, iters=10, seed=5):
    rng = np.random.default_rng(seed)
    hmax = np.uint64(1) << np.uint64(58)
    dbh = rng.integers(0, int(hmax), size=(G, n), dtype=np.uint64)
    dbh.sort(axis=1)
    dbh = dbh.reshape(-1)
    dbo = (np.arange(G + 1, dtype=np.uint64) * np.uint64(n))
    t0 = time.perf_counter()
    table = hip.upload_table(dbh, dbo)
    t_up = time.perf_counter() - t0
    member = rng.choice(dbh, size=Q // 2, replace=False) if Q // 2 <= len(dbh) else dbh
    other = rng.integers(0, int(hmax), size=Q - len(member), dtype=np.uint64)
    qh = np.unique(np.concatenate([member, other]))
    qc = rng.integers(1, 4, size=len(qh)).astype(np.uint32)
    d_h, d_c = hip.array(qh), hip.array(qc)
    sk = hip.sketch_from_pairs_dev(d_h.ptr, d_c.ptr, len(qh), 21)
    d_hits, d_sizes = hip.empty(G, np.uint32), hip.empty(G, np.uint32)
    hip.containment_dev(sk, table, 2, d_hits.ptr, d_sizes.ptr)
    hip.sync()
    hits = d_hits.download()
    # check against numpy on a sample of genomes
    good = set(qh[qc >= 2].tolist()) if len(qh) < 3_000_000 else None
    if good is not None:
        for g in rng.integers(0, G, size=5):
            want = sum(1 for h in dbh[g * n:(g + 1) * n].tolist() if h in good)
            assert want == hits[g], (g, want, hits[g])
    hip.prof_reset()
    hip.prof_enable(True)
    for _ in range(iters):
        hip.containment_dev(sk, table, 2, d_hits.ptr, d_sizes.ptr)
    hip.sync()
    hip.prof_enable(False)
    t = prof(hip, ["contain_index", "containment"])
    algo = G * n * 8 + len(qh) * 12
    res = {"G": G, "n": n, "Q": int(len(qh)), "upload_s": round(t_up, 3), "containment_ms": t.get("containment"),
           "algo_bytes": algo, "GBps": algo / (t["containment"] * 1e-3) / 1e9}
    sk.free(); table.free(); d_h.free(); d_c.free(); d_hits.free(); d_sizes.free()
    return res


def stage_c(hip, R, G, iters=10, seed=7):
    rng = np.random.default_rng(seed)
    src = rng.integers(1, G + 1, size=R)
    recs = synth.make_alignment_records(src, G + 1, seed=seed)
    ref2tax = np.arange(G + 1, dtype=np.uint32)
    T = G + 1
    d_recs, d_r2t = hip.array(recs), hip.array(ref2tax)
    d_acc = hip.empty(3 * T + 2, np.uint64)
    names = ["profile_map", "profile_pass"]

    def run():
        d_acc.memset(0)
        sh = hip.profile_begin_dev(d_recs.ptr, len(recs), False, d_r2t.ptr, T, T, 0.5)
        sh.commit(True, True, 0, d_acc.ptr, d_acc.ptr + 8 * T, d_acc.ptr + 16 * T, d_acc.ptr + 24 * T)
        return sh

    sh = run()
    hip.sync()
    sh.free()
    hip.prof_reset()
    hip.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        sh = run()
        sh.free()
    hip.sync()
    wall = (time.perf_counter() - t0) / iters
    hip.prof_enable(False)
    t = prof(hip, names)
    tot = sum(t.values())
    algo = len(recs) * 16
    res = {"R": R, "T": T, "records": int(len(recs)), "kernels_ms": {k: round(v, 4) for k, v in t.items()},
           "sum_ms": tot, "wall_ms": wall * 1e3, "algo_bytes": algo, "GBps": algo / (tot * 1e-3) / 1e9}
    d_recs.free(); d_r2t.free(); d_acc.free()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--big", action="store_true", help="also the 200k-genome table and 100M reads on one GPU")
    a = ap.parse_args()
    hip = Hip.get(0)
    res = {"device": hip.device_name(), "stage_b": [], "stage_c": []}
    for G, Q in [(1000, 700_000), (10_000, 5_000_000), (25_000, 5_000_000)] + ([(200_000, 20_000_000)] if a.big else []):
        r = stage_b(hip, G, 1000, Q)
        print("B", json.dumps(r), flush=True)
        res["stage_b"].append(r)
    for R, G in [(1_000_000, 1000), (10_000_000, 10_000), (12_500_000, 2000)] + ([(100_000_000, 10_000)] if a.big else []):
        r = stage_c(hip, R, G)
        print("C", json.dumps(r), flush=True)
        res["stage_c"].append(r)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
