#!/usr/bin/env python3
"""A `.fq.gz` (one gzip member, and BGZF) through the library with the DEVICE inflater (mg_inflate.hip) and with the host
inflaters of round 4, as the whole file -> HBM -> parse -> hash pipeline (mg_sketch_stream_add_file), beside the plain file;
per-kernel times of the device inflater from the library's own events.

    python tools/inflate_probe.py [reads] [realistic_qualities 0|1]
"""
import json
import os
import struct
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from metalign_amd import _hip, synth  # noqa: E402
import bench_cli  # noqa: E402


from bench_cli import parallel_gzip, parallel_bgzf  # noqa: E402,F401


_WARM = {}


def warm_clocks(hip, seconds=0.15):
    """The device idles between the probe's runs and takes tens of milliseconds of work to come back to full clocks: genome sketching
    in a loop right before every timed run, so that a run measures the kernels and not the clock governor."""
    if not _WARM:
        _WARM["g"] = synth.make_genomes(200, 50_000)
    gb, go = _WARM["g"]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        hip.sketch_genomes(gb, go, 31, 1000)


KERNELS = ("k_find_block_starts", "k_inflate", "k_inflate_bgzf", "k_window_chain", "k_resolve_text", "k_crc_segments")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    realistic = len(sys.argv) > 2 and sys.argv[2] == "1"
    td = tempfile.mkdtemp(prefix="mg_gz_")
    gb, go = synth.make_genomes(500, 50_000)
    rb, ro, src = synth.make_reads(gb, go, n, npresent=50)
    fq = os.path.join(td, "reads.fq")
    nbytes = bench_cli.write_fastq(fq, rb, n)
    text = np.fromfile(fq, dtype=np.uint8)
    if realistic:  # binned qualities as a sequencer writes them, instead of the constant 'I'
        rng = np.random.default_rng(7)
        rec = text.reshape(n, -1)
        q = rng.choice(np.frombuffer(b"FFFFFFFF:,#", dtype=np.uint8), size=(n, 150))
        rec[:, 15 + 150:15 + 300] = q
        text.tofile(fq)
    threads = min(os.cpu_count() or 8, 64)
    t0 = time.perf_counter()
    gz = parallel_gzip(text, threads=threads)
    bz = parallel_bgzf(text, threads=threads)
    open(fq + ".gz", "wb").write(gz)
    open(fq + ".bgzf.gz", "wb").write(bz)
    out = {"reads": n, "fastq_bytes": nbytes, "gz_bytes": len(gz), "bgzf_bytes": len(bz), "compress_s": time.perf_counter() - t0, "host_cores": os.cpu_count(),
           "qualities": "binned, random" if realistic else "constant"}
    assert zlib.decompress(gz[: 50 << 20] if False else gz, 47)[:1000] == text[:1000].tobytes()
    hip = _hip.Hip.get(0)
    if os.environ.get("MG_INFLATE_TRACE"):  # every stage's jobs, the jobs decoded twice and the holes of the chain on stderr
        _hip.debug_set("inflate_trace", 1)
    if os.environ.get("MG_PROBE_STAGE_MB"):  # compressed bytes per stage instead of the device's round of jobs
        hip.inflate_config(stage_bytes=int(os.environ["MG_PROBE_STAGE_MB"]) << 20)
    if os.environ.get("MG_PROBE_CHUNK_KB"):
        hip.inflate_config(chunk_bytes=int(os.environ["MG_PROBE_CHUNK_KB"]) << 10)
    k = 51
    dbh, dbo = hip.sketch_genomes(gb, go, k, 1000)
    hmax = int(dbh.max())
    filt = hip.filter_build(dbh)
    out["pipeline"] = {}
    cases = (("plain", fq, None), ("gzip_device", fq + ".gz", 1), ("bgzf_device", fq + ".bgzf.gz", 1), ("gzip_host_threads", fq + ".gz", 0), ("bgzf_host_threads", fq + ".bgzf.gz", 0))
    for name, path, on in cases:
        if on is not None:
            hip.inflate_config(on=on)
        best, sizes, stats, kern = None, None, None, None
        for rep in range(3):
            warm_clocks(hip)
            st = hip.sketch_stream([k], [hmax], 0, [filt], nbytes // 2)
            hip.inflate_stats(reset=True)
            prof = rep == 2 and on == 1
            if prof:
                hip.prof_enable(True)
                hip.prof_reset()
            t0 = time.perf_counter()
            st.add_file(path, "fastq")
            sks = st.finish()
            for sk in sks:
                sk.resolve()
            dt = time.perf_counter() - t0
            if prof:
                kern = {kn: dict(zip(("launches", "ms"), hip.prof_get(kn))) for kn in KERNELS}
                hip.prof_enable(False)
            sizes = [sk.size for sk in sks]
            nreads = st.nreads
            for sk in sks:
                sk.free()
            st.free()
            if not prof and (best is None or dt < best):
                best, stats = dt, hip.inflate_stats()
        assert nreads == n, (name, nreads)
        out["pipeline"][name] = {"seconds": best, "reads_per_s": n / best, "text_GBs": nbytes / best / 1e9, "sketch_sizes": sizes}
        if on == 1:
            out["pipeline"][name]["inflate_stats"] = stats
            out["pipeline"][name]["kernels_ms_profiled_run"] = kern
    hip.inflate_config(on=1)
    ref = out["pipeline"]["plain"]["sketch_sizes"]
    assert all(v["sketch_sizes"] == ref for v in out["pipeline"].values()), "the sketches differ"
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
