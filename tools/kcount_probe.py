"""Stage A by k-mer identity on bench.py's configs[2] workload (10M synthetic 150 bp reads against the 10k-genome table):
k_count_kmers alone (HIP events around N launches), the hash path's read sketch beside it, stage B from the counters, and the
columns of both held to each other.  `--knob key=value` sets library knobs (kc_wg_per_cu ...).

    python tools/kcount_probe.py [--reads 10000000] [--genomes 10000] [--ks 21,31,51] [--reps 10]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=10_000_000)
    ap.add_argument("--genomes", type=int, default=10_000)
    ap.add_argument("--genome_len", type=int, default=50_000)
    ap.add_argument("--ks", default="21,31,51")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--knob", action="append", default=[])
    ap.add_argument("--no_hash_path", action="store_true")
    ap.add_argument("--cs", type=int, default=3, help="the counters' saturation (kmc -cs); 0 = none")
    args = ap.parse_args()
    from metalign_amd import _hip, synth
    from metalign_amd._hip import Hip
    hip = Hip.get(0)
    for kv in args.knob:
        k, v = kv.split("=")
        _hip.debug_set(k, int(v))
    ks = [int(x) for x in args.ks.split(",")]
    hip.count_saturation(args.cs)
    t0 = time.time()
    gb, go = synth.make_genomes(args.genomes, args.genome_len)
    rb, ro, _ = synth.make_reads(gb, go, args.reads, npresent=max(50, args.genomes // 20))
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], 1000)
    table = hip.refdb_build(h, khi, klo, o, ks)
    t1 = time.time()
    table.index_kmers()
    hip.sync()
    t2 = time.time()
    d_b = hip.array(np.concatenate([rb, np.zeros(64, np.uint8)]))
    d_o = hip.array(ro)
    n, nb = args.reads, int(ro[-1])
    kc = table.kmer_counts()
    out = dict(reads=n, genomes=args.genomes, ks=ks, build_s=round(t1 - t0, 1), index_s=round(t2 - t1, 3), distinct_kmers=table.distinct_kmers)

    # HIP events through the library are record / synchronize only: time with the host clock around a synchronised batch
    def wall(fn, reps):
        fn()
        hip.sync()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        hip.sync()
        return (time.perf_counter() - t) * 1e3 / reps

    def count():
        kc.reset()
        kc.add_dev(d_b.ptr, d_o.ptr, n, nb)
    out["count_kmers_ms"] = round(wall(count, args.reps), 3)
    out["stats"] = kc.stats()
    g, nk = table.ngenomes, len(ks)
    d = hip.empty(max(2 * g * nk, 1), np.uint32)
    hp = [d.ptr + 4 * (2 * ki * g) for ki in range(nk)]
    sp = [d.ptr + 4 * ((2 * ki + 1) * g) for ki in range(nk)]
    out["stage_b_counts_ms"] = round(wall(lambda: hip.refpipe_containment_counts_dev(kc, table, 2, hp, sp), args.reps), 3)
    cols = d.download()[: 2 * g * nk].reshape(nk, 2, g).copy()
    out["genomes_with_hits"] = [int((cols[ki, 0] > 0).sum()) for ki in range(nk)]
    if not args.no_hash_path:
        filt = hip.filter_build(h)
        sk = [None]

        def sketch():
            if sk[0] is not None:
                sk[0].free()
            sk[0] = hip.sketch_reads_dev_async(d_b.ptr, d_o.ptr, n, ks[-1], table.max_hash, 0, filt)
        out["hash_path_sketch_ms"] = round(wall(sketch, max(args.reps // 2, 2)), 3)
        out["stage_b_sketch_ms"] = round(wall(lambda: hip.refpipe_containment_dev(sk[0], table, 2, hp, sp), args.reps), 3)
        cols_h = d.download()[: 2 * g * nk].reshape(nk, 2, g)
        out["columns_equal_hash_path"] = bool(np.array_equal(cols, cols_h))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
