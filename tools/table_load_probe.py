#!/usr/bin/env python3
"""What select_main spends before the reads stream: the stored reference-pipeline table (formats.py, version 3) opened, mapped, uploaded
(mg_refdb_upload), its membership filter uploaded — step by step, files in the page cache.

    python tools/table_load_probe.py [genomes] [genome_len]
"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from metalign_amd import _hip, formats, synth  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
    ks = [21, 31, 51]
    hip = _hip.Hip.get(0)
    gb, go = synth.make_genomes(G, L)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], 1000)
    t = hip.refdb_build(h, khi, klo, o, ks)
    arrays = t.download(kmers=False)
    t.free()
    f = hip.filter_build(arrays["pair_hash"])
    td = tempfile.mkdtemp(prefix="mg_tab_")
    formats.write_refpipe_table(td, ["g%d" % i for i in range(G)], 1000, arrays, f.download())
    f.free()
    nbytes = sum(os.path.getsize(os.path.join(td, x)) for x in os.listdir(td))
    for rep in range(4):
        t0 = time.perf_counter()
        table = formats.SketchTable(td)
        t1 = time.perf_counter()
        a = table.refpipe_arrays()
        t2 = time.perf_counter()
        ref = hip.refdb_upload(a["ks"], a["ngenomes"], a["pair_hash"], a["pair_gen"], a["gsize"], a["max_hash"], a["small"])
        hip.sync()
        t3 = time.perf_counter()
        bits = table.filter_bits(ks[-1])
        flt = hip.filter_from_bits(bits)
        hip.sync()
        t4 = time.perf_counter()
        print("run %d: %d MB on disk | open %.1f ms, map %.1f ms, mg_refdb_upload %.1f ms (%.1f GB/s), filter %.1f ms | total %.1f ms"
              % (rep, nbytes >> 20, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), nbytes / (t3 - t2) / 1e9, 1e3 * (t4 - t3), 1e3 * (t4 - t0)), flush=True)
        ref.free()
        flt.free()


if __name__ == "__main__":
    main()
