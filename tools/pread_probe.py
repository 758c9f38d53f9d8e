"""How fast can T threads read a file out of the page cache (pread into private 32 MB buffers, nothing else)?  The ceiling the
streaming readers have on this host, against the PCIe link's ~50 GB/s.   python tools/pread_probe.py [GB]"""
import os, sys, tempfile, threading, time
import numpy as np
gb = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
td = tempfile.mkdtemp(prefix="mg_pp_")
path = os.path.join(td, "f.bin")
blk = np.random.default_rng(1).integers(32, 127, size=64 << 20, dtype=np.uint8).tobytes()
with open(path, "wb") as fh:
    for _ in range(int(gb * 1024 / 64)):
        fh.write(blk)
size = os.path.getsize(path)
open(path, "rb").read(1 << 20)
CH = 32 << 20
for T in (1, 2, 4, 8, 16, 32, 64):
    best = None
    for rep in range(3):
        nxt = [0]
        lock = threading.Lock()
        def work():
            fd = os.open(path, os.O_RDONLY)
            buf = bytearray(CH)
            while True:
                with lock:
                    i = nxt[0]; nxt[0] += 1
                if i * CH >= size: break
                os.preadv(fd, [buf], i * CH)
            os.close(fd)
        ths = [threading.Thread(target=work) for _ in range(T)]
        t0 = time.perf_counter()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    print("%2d threads: %.3f s = %.1f GB/s" % (T, best, size / best / 1e9), flush=True)
import shutil; shutil.rmtree(td, ignore_errors=True)
