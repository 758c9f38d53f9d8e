"""SAM file -> alignment records through mg_sam_stream_file for several chunk sizes / reader-thread counts: seconds, GB/s of
text, and the device time of the tokeniser's kernels (mg_prof_*: ingest_lines = newline count + marks, ingest_sam = parse, scan,
list, emit) — is the stream bound by the link, by the readers or by the tokeniser?
python tools/sam_stream_probe.py [nreads] [ngenomes]"""
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
n -= n % 4
hip = Hip.get(0)
gb, go = synth.make_genomes(G, 5_000)
rb, ro, src = synth.make_reads(gb, go, n, npresent=max(40, G // 20))
td = tempfile.mkdtemp(prefix="mg_ssp_")
sam = os.path.join(td, "aln.sam")
nbytes, nlines = bench_cli.write_sam(sam, rb[: n * 150], np.asarray(src[:n], dtype=np.int64), n, G)
index = hip.acc_index([bench_cli.ACC % g for g in range(G)])
print("aln.sam: %.2f GB, %d lines" % (nbytes / 1e9, nlines), flush=True)


def once(chunk, threads, prof=False):
    if prof:
        hip.prof_reset(); hip.prof_enable(True)
    t0 = time.perf_counter()
    b = hip.sam_stream_file(sam, index, chunk_bytes=chunk, nthreads=threads)
    dt = time.perf_counter() - t0
    cnt = b.count
    b.free()
    extra = ""
    if prof:
        hip.sync(); hip.prof_enable(False)
        extra = "  device: " + ", ".join("%s %.1f ms / %d" % (k, hip.prof_get(k)[1], hip.prof_get(k)[0]) for k in ("ingest_lines", "ingest_sam"))
    return dt, cnt, extra


once(32 << 20, 8)
for chunk_mb in (16, 32, 64, 128):
    for threads in (4, 8, 16):
        cold, cnt, _ = once(chunk_mb << 20, threads)
        warm = min(once(chunk_mb << 20, threads)[0] for _ in range(3))
        _, _, extra = once(chunk_mb << 20, threads, prof=True)
        print("chunk %3d MB, %2d readers: cold %.3f s, warm %.3f s = %.1f GB/s = %.2e lines/s  (%d records)%s"
              % (chunk_mb, threads, cold, warm, nbytes / warm / 1e9, nlines / warm, cnt, extra), flush=True)
import shutil
shutil.rmtree(td, ignore_errors=True)
