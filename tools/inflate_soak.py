#!/usr/bin/env python3
"""Random gzip streams through the device inflater (mg_inflate_dev) against zlib: texts, FASTQ-like records, binary noise, long runs,
mixtures; levels 1-9, several members, Z_SYNC / Z_FULL flushes, Z_FIXED / Z_HUFFMAN_ONLY / Z_RLE strategies, small and large chunk / stage
settings, both decoders.  python tools/inflate_soak.py [streams] [seed] [corrupt]
(corrupt: every stream damaged — bits flipped, cut — and zlib's verdict expected: an error, or the same text)."""
import os
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import _hip  # noqa: E402


def payload(rng, n):
    kind = int(rng.integers(0, 6))
    if kind == 0:
        return rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
    if kind == 1:
        return bytes(rng.choice(np.frombuffer(b"ACGTN\n", np.uint8), size=n, p=[.24, .24, .24, .24, .02, .02]))
    if kind == 2:
        recs = []
        while sum(map(len, recs)) < n:
            L = int(rng.integers(50, 251))
            seq = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=L))
            q = bytes(rng.choice(np.frombuffer(b"FFFFFF:,#", np.uint8), size=L))
            recs.append(b"@r%d/%d\n%s\n+\n%s\n" % (len(recs), int(rng.integers(1, 3)), seq, q))
        return b"".join(recs)[:n]
    if kind == 3:
        return (bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 70000)) + b"xyz" * int(rng.integers(1, 5000)))[:n] * 2
    if kind == 4:
        words = [bytes(rng.integers(97, 123, size=int(rng.integers(2, 12)), dtype=np.uint8)) for _ in range(300)]
        return b" ".join(words[int(i)] for i in rng.integers(0, 300, size=n // 6))[:n]
    a = payload(rng, n // 3)
    return a + rng.integers(0, 4, size=n // 3, dtype=np.uint8).tobytes() + a[::-1]


def compress(rng, data):
    out = b""
    nmem = int(rng.integers(1, 4))
    cuts = sorted(int(x) for x in rng.integers(0, len(data) + 1, size=nmem - 1))
    for a, b in zip([0] + cuts, cuts + [len(data)]):
        strat = [zlib.Z_DEFAULT_STRATEGY] * 4 + [zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED]
        c = zlib.compressobj(int(rng.integers(1, 10)), zlib.DEFLATED, 31, int(rng.integers(1, 10)), strat[int(rng.integers(0, len(strat)))])
        piece = data[a:b]
        at = 0
        while at < len(piece):
            step = int(rng.integers(1, max(2, len(piece))))
            out += c.compress(piece[at:at + step])
            at += step
            if rng.random() < 0.2:
                out += c.flush(zlib.Z_SYNC_FLUSH if rng.random() < 0.5 else zlib.Z_FULL_FLUSH)
        out += c.flush()
    return out


SIZES = ([0, 1, 100, 5000, 70_000, 400_000, 3_000_000, 12_000_000], [.02, .03, .1, .15, .2, .25, .2, .05])


def soak(hip, nstreams, seed, corrupt, sizes=SIZES):
    rng = np.random.default_rng(seed)
    total = refused = 0
    for i in range(nstreams):
        n = int(rng.choice(sizes[0], p=sizes[1]))
        data = payload(rng, n) if n else b""
        gz = compress(rng, data)
        hip.inflate_config(chunk_bytes=int(rng.choice([4 << 10, 16 << 10, 32 << 10, 100_000])), stage_bytes=int(rng.choice([-1, -1, 300_000, 2 << 20])),
                           ratio=int(rng.choice([10, 10, 2, 40])), lane_jobs=0 if rng.random() < 0.2 else 1 << 40)
        if corrupt and len(gz) > 30:  # a damaged stream: zlib's verdict is the expected one (an error, or — a flipped header byte — the same text)
            b = bytearray(gz)
            for _ in range(int(rng.integers(1, 4))):
                if rng.random() < 0.3:
                    del b[int(rng.integers(len(b) // 2, len(b))):]  # cut
                else:
                    b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
            gz = bytes(b)
            try:
                d = zlib.decompressobj(47)
                want, rest = d.decompress(gz), d.unused_data
                while rest and b"\x1f\x8b\x08".startswith(rest[:3]) or rest[:3] == b"\x1f\x8b\x08":  # further members, as gzip reads them:
                    # what follows a member is one if it starts like one — also when the file ends inside the three bytes that say so
                    if len(rest) < 3:
                        raise zlib.error("ends inside a member header")
                    d = zlib.decompressobj(31)
                    want += d.decompress(rest)
                    if not d.eof:
                        raise zlib.error("truncated member")
                    rest = d.unused_data
                if not d.eof:
                    raise zlib.error("truncated")
            except zlib.error:
                want = None
            try:
                got = hip.inflate(gz)
            except OSError:
                got = None
            assert got == want, "damaged stream %d: zlib %s, device %s" % (i, "refuses" if want is None else "%d bytes" % len(want),
                                                                         "refuses" if got is None else "%d bytes" % len(got))
            refused += want is None
            continue
        got = hip.inflate(gz)
        assert got == data, "stream %d: %d bytes in, %d out, %d expected" % (i, len(gz), len(got), len(data))
        total += len(data)
    hip.inflate_config(chunk_bytes=32 << 10, stage_bytes=-1, ratio=10, lane_jobs=1 << 40)
    return total, refused


def main():
    nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    corrupt = len(sys.argv) > 3 and sys.argv[3] == "corrupt"
    total, refused = soak(_hip.Hip.get(0), nstreams, seed, corrupt)
    if corrupt:
        print("ok: %d damaged streams, %d refused by both, the rest read alike" % (nstreams, refused))
    else:
        print("ok: %d streams, %.1f MB of text, all equal to what zlib compressed" % (nstreams, total / 1e6))


if __name__ == "__main__":
    main()
