#!/usr/bin/env python3
"""A plain `.fq.gz` through the library: zlib on one thread (rounds 2-3) against the parallel inflater (mg_pgzip.hip), as bare
host inflating (mg_gunzip_*) and as the whole file -> HBM -> parse -> hash pipeline (mg_sketch_stream_add_file) beside the same
reads as plain text and as BGZF.

    python tools/gzip_probe.py [reads] [threads,...]
"""
import gzip
import json
import os
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from metalign_amd import _hip, synth  # noqa: E402
import bench_cli  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    threads = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8, 16, 32, 64]
    td = tempfile.mkdtemp(prefix="mg_gz_")
    gb, go = synth.make_genomes(500, 50_000)
    rb, ro, src = synth.make_reads(gb, go, n, npresent=50)
    fq = os.path.join(td, "reads.fq")
    nbytes = bench_cli.write_fastq(fq, rb, n)
    t0 = time.perf_counter()
    subprocess.check_call("gzip -6 -k -c %s > %s.gz" % (fq, fq), shell=True) if not os.system("which pigz > /dev/null 2>&1") else None
    if not os.path.exists(fq + ".gz"):
        with open(fq, "rb") as fi, gzip.open(fq + ".gz", "wb", 6) as fo:
            while True:
                b = fi.read(64 << 20)
                if not b:
                    break
                fo.write(b)
    out = {"reads": n, "fastq_bytes": nbytes, "gz_bytes": os.path.getsize(fq + ".gz"), "compress_s": time.perf_counter() - t0, "host_cores": os.cpu_count()}
    blob = open(fq + ".gz", "rb").read()
    t0 = time.perf_counter()
    want = zlib.decompress(blob, 47)
    out["zlib_one_thread"] = {"seconds": time.perf_counter() - t0, "text_GBs": nbytes / (time.perf_counter() - t0) / 1e9}
    assert len(want) == nbytes
    out["mg_gunzip"] = {}
    for th in threads:
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            got = _hip.gunzip_file(fq + ".gz", nthreads=th)
            dt = time.perf_counter() - t0
            assert got == want
            best = dt if best is None else min(best, dt)
        out["mg_gunzip"][str(th)] = {"seconds": best, "text_GBs": nbytes / best / 1e9}
    # the streaming pipeline: file -> page-locked slots -> HBM -> parser -> one set of counting tables
    hip = _hip.Hip.get(0)
    k = 51
    dbh, dbo = hip.sketch_genomes(gb, go, k, 1000)
    hmax = int(dbh.max())
    filt = hip.filter_build(dbh)
    out["pipeline"] = {}
    for name, path, env in (("plain", fq, {}), ("gzip_parallel", fq + ".gz", {}), ("gzip_zlib_one_thread", fq + ".gz", {"gzip_threads": 1})):
        hip.inflate_config(on=0)  # (this probe is about the HOST inflaters; tools/inflate_probe.py has the device one beside them)
        for kk, v in env.items():
            _hip.debug_set(kk, v)
        best, sizes = None, None
        for rep in range(2):
            st = hip.sketch_stream([k], [hmax], 0, [filt], nbytes // 2)
            t0 = time.perf_counter()
            st.add_file(path, "fastq")
            sks = st.finish()
            for sk in sks:
                sk.resolve()
            dt = time.perf_counter() - t0
            sizes = [sk.size for sk in sks]
            for sk in sks:
                sk.free()
            st.free()
            best = dt if best is None else min(best, dt)
        for kk in env:
            _hip.debug_set(kk, 0)
        out["pipeline"][name] = {"seconds": best, "reads_per_s": n / best, "text_GBs": nbytes / best / 1e9, "sketch_sizes": sizes}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
