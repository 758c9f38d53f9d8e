#!/usr/bin/env python3
"""The reference pipeline (stage A at the largest k only + prefix columns on the table side) beside a sketch per k, on
bench.py's workload shapes: ms per pipelined pass, per-kernel times, and every column of a sample against the oracle.

    python tools/refpipe_probe.py [reads] [genomes] [genome_len] [ks] [check_reads]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from metalign_amd import distributed as mgd, synth  # noqa: E402
from metalign_amd._hip import Hip  # noqa: E402


def main():
    nreads = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
    glen = int(sys.argv[3]) if len(sys.argv) > 3 else 50_000
    ks = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "21,31,51").split(",")]
    ncheck = int(sys.argv[5]) if len(sys.argv) > 5 else 200_000
    steps = int(os.environ.get("STEPS", 20))
    hip = Hip.get(0)
    gb, go = synth.make_genomes(G, glen)
    rb, ro, src = synth.make_reads(gb, go, nreads, npresent=max(50, G // 20))
    recs = synth.make_alignment_records(src + 1, G + 1)
    ref2tax = np.arange(G + 1, dtype=np.uint32)
    out = {"workload": dict(reads=nreads, genomes=G, genome_len=glen, ks=ks)}

    def timed(job):
        job.run(3)
        hip.sync()
        hip.prof_reset(); hip.prof_enable(True); hip.prof_only("sketch_reads")
        t0 = time.perf_counter()
        res = job.run(steps)
        hip.sync()
        dt = (time.perf_counter() - t0) / steps
        n, t = hip.prof_get("sketch_reads")
        hip.prof_enable(False)
        # per-kernel table on one stream
        hip.prof_reset(); hip.prof_enable(True); hip.stage_c_side_stream(False)
        for _ in range(3):
            job.step()
        hip.sync()
        hip.stage_c_side_stream(True)
        kern = {}
        for name in ("table_clear", "sketch_reads", "bucket_sort", "bucket_pack", "contain_index", "containment", "refpipe_count", "profile_pass"):
            c, ms = hip.prof_get(name)
            if c:
                kern[name] = round(ms / 3, 4)
        hip.prof_enable(False)
        return dict(ms_per_pass=1e3 * dt, reads_per_s=nreads / dt, k1_avg_launch_ms=t / max(n, 1), kernels_ms=kern,
                    sketch_sizes=res["sketch_sizes"]), res

    if os.environ.get("SKIP_PER_K") != "1":
        hip.set_hash_mode(0)
        tabs = [hip.sketch_genomes(gb, go, k, 1000) for k in ks]
        job = mgd.ShardJob(hip, None, 0, 1, k=ks)
        job.load(rb, ro, recs, ref2tax, [t[0] for t in tabs], [t[1] for t in tabs], ntax=G + 1)
        out["sketch_per_k_mode0"], _ = timed(job)
        del job, tabs
        hip.mem_trim()
    for mode in (0, 1):
        hip.set_hash_mode(mode)
        t0 = time.perf_counter()
        h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], 1000)
        t1 = time.perf_counter()
        table = hip.refdb_build(h, khi, klo, o, ks)
        t2 = time.perf_counter()
        arrays = table.download(kmers=False)
        job = mgd.ShardJob(hip, None, 0, 1, k=ks, definition="reference_pipeline")
        job.load(rb, ro, recs, ref2tax, arrays, ntax=G + 1, reftable=table)
        r, res = timed(job)
        r["table_build_s"] = dict(sketch_genomes_kmers=t1 - t0, refdb_build=t2 - t1)
        if ncheck:
            import oracle
            oracle.build()
            oracle.set_hash_mode(mode)
            want = oracle.refpipe_build(h, khi, klo, o, ks)
            sub = mgd.ShardJob(hip, None, 0, 1, k=ks, definition="reference_pipeline")
            nrec = int(np.searchsorted(np.cumsum(recs["ref_new"] >> 31), ncheck, side="right"))
            sub.load(rb[: int(ro[ncheck])], ro[: ncheck + 1], recs[:nrec], ref2tax, arrays, ntax=G + 1, reftable=table)
            got = sub.step()
            qh, qc, _, _ = oracle.sketch_reads(rb[: int(ro[ncheck])], ro[: ncheck + 1], ks[-1], hmax=int(h.max()))
            whits, wsizes = oracle.refpipe_containment(qh, qc, 2, want)
            r["check"] = dict(reads=ncheck, hits_equal=bool(np.array_equal(got["hits_k"], whits)),
                              sizes_equal=bool(np.array_equal(got["sizes_k"], wsizes)),
                              genomes_hit=[int((whits[i] > 0).sum()) for i in range(len(ks))])
            oracle.set_hash_mode(0)
            del sub
        out["reference_pipeline_mode%d" % mode] = r
        del job
        table.free()
        hip.mem_trim()
    hip.set_hash_mode(0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
