"""Reads file -> read sketches through mg_sketch_stream_add_file for several chunk sizes / reader-thread counts:
cold (the page-locked slots of that size are allocated in the call) and warm seconds, GB/s of text.
python tools/stream_probe.py [nreads] [ngenomes]"""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_cli  # noqa: E402
from metalign_amd import synth  # noqa: E402
from metalign_amd._hip import Hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
ks = [int(x) for x in os.environ.get("MG_PROBE_KS", "21,31,51").split(",")]  # (the reference pipeline sketches the largest k only)
hip = Hip.get(0)
gb, go = synth.make_genomes(G, 50_000)
rb, ro, src = synth.make_reads(gb, go, n, npresent=max(40, G // 20))
tabs = [hip.sketch_genomes(gb, go, k, 1000)[0] for k in ks]
hmaxs = [int(t.max()) for t in tabs]
filts = [hip.filter_build(t) for t in tabs]
td = tempfile.mkdtemp(prefix="mg_sp_")
fq = os.path.join(td, "reads.fq")
nbytes = bench_cli.write_fastq(fq, rb, n)
open(fq, "rb").read(1 << 20)
print("reads.fq: %.2f GB" % (nbytes / 1e9), flush=True)


def once(chunk, threads):
    st = hip.sketch_stream(ks, hmaxs, 0, filts, expect_bases=rb.size)
    t0 = time.perf_counter()
    st.add_file(fq, "fastq", chunk_bytes=chunk, nthreads=threads)
    sks = st.finish()
    for sk in sks:
        sk.resolve()
    dt = time.perf_counter() - t0
    sizes = [sk.size for sk in sks]
    for sk in sks:
        sk.free()
    st.free()
    return dt, sizes


once(64 << 20, 8)  # distinct-count hint, clocks
if os.environ.get("MG_PROBE_AB"):  # a knob of the library off / on, alternating, at the pipeline's default chunk size and readers
    from metalign_amd import _hip as _hipmod
    for rep in range(4):
        for v in (0, 1):
            _hipmod.debug_set(os.environ["MG_PROBE_AB"], v)
            warm = min(once(0, 0)[0] for _ in range(3))
            print("%s = %d: warm %.4f s = %.1f GB/s" % (os.environ["MG_PROBE_AB"], v, warm, nbytes / warm / 1e9), flush=True)
    _hipmod.debug_set(os.environ["MG_PROBE_AB"], 0)
    sys.exit(0)
for chunk_mb in (8, 16, 32, 64, 128):
    for threads in (4, 8, 16, 32):
        cold, s1 = once(chunk_mb << 20, threads)
        warm = min(once(chunk_mb << 20, threads)[0] for _ in range(3))
        print("chunk %3d MB, %2d readers: cold %.3f s, warm %.3f s = %.1f GB/s = %.2e reads/s   %s"
              % (chunk_mb, threads, cold, warm, nbytes / warm / 1e9, n / warm, s1), flush=True)
import shutil
shutil.rmtree(td, ignore_errors=True)
