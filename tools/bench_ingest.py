"""Throughput of the device-side ingest kernels on text resident in HBM (ingest).
python tools/bench_ingest.py [nreads]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import synth  # noqa: E402
from metalign_amd._hip import Hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
hip = Hip.get(0)
gb, go = synth.make_genomes(200, 50_000)
rb, ro, src = synth.make_reads(gb, go, n, npresent=50)
seqs = rb.reshape(n, 150)
# FASTQ text: "@r<i>\n<seq>\n+\n<qual>\n"
t0 = time.perf_counter()
names = np.char.add(np.char.add("@r", np.arange(n).astype(str)), "\n").astype("S")
qual = b"I" * 150 + b"\n"
parts = []
for i in range(n):
    parts.append(names[i])
    parts.append(seqs[i].tobytes())
    parts.append(b"\n+\n")
    parts.append(qual)
fq = b"".join(parts)
# SAM text: one line per read + 25 % secondaries
accs = ["NZ_SYN%06d.1" % g for g in range(200)]
lines = []
for i in range(n):
    s = seqs[i].tobytes().decode()
    lines.append("r%d\t%d\t%s\t1000\t60\t150M\t*\t0\t0\t%s\t%s\tNM:i:1\n" % (i, 16 * (i & 1), accs[src[i]], s, "I" * 150))
    if i % 4 == 0:
        lines.append("r%d\t256\t%s\t1000\t0\t140M10S\t*\t0\t0\t*\t*\tNM:i:5\n" % (i, accs[(src[i] + 1) % 200]))
sam = "".join(lines).encode()
print("text built in %.1f s: fastq %.1f MB, sam %.1f MB" % (time.perf_counter() - t0, len(fq) / 1e6, len(sam) / 1e6))
d_fq = hip.array(np.frombuffer(fq, np.uint8))
d_sam = hip.array(np.frombuffer(sam, np.uint8))
idx = hip.acc_index(["Unmapped"] + accs)
for name, fn, nbytes in (("fastq", lambda: hip.parse_reads_dev(d_fq.ptr, len(fq), "fastq").free(), len(fq)),
                         ("sam", lambda: hip.sam_tokenize_dev(d_sam.ptr, len(sam), idx), len(sam))):
    for _ in range(3):
        fn()
    hip.sync()
    hip.prof_reset(); hip.prof_enable(True)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        fn()
    hip.sync()
    dt = (time.perf_counter() - t0) / reps
    hip.prof_enable(False)
    ks = {k: hip.prof_get(k) for k in ("ingest_lines", "ingest_reads", "ingest_sam")}
    print("%-6s %.3f ms per call  %.1f GB/s of text  %.2e reads/s   kernels(ms): %s" % (
        name, dt * 1e3, nbytes / dt / 1e9, n / dt, {k: round(v[1] / max(v[0], 1), 3) for k, v in ks.items() if v[0]}))
