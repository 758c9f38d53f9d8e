"""CPU: the library's parallel gzip inflater (mg_pgzip.hip, mg_gunzip_*: one deflate stream entered in the middle by many
threads) against Python's zlib on the same files.  Host code of the library: runs without a GPU."""
import gzip
import os
import zlib

import numpy as np
import pytest

from metalign_amd import _hip


def _fastq(rng, nreads, readlen=150):
    seqs = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(nreads, readlen))
    qual = rng.integers(35, 74, size=(nreads, readlen)).astype(np.uint8)
    out = []
    for i in range(nreads):
        out.append(b"@read%d/1 len=%d\n" % (i, readlen) + seqs[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n")
    return b"".join(out)


@pytest.fixture(autouse=True)
def _knobs_back():
    yield
    _hip.debug_set(None)


@pytest.fixture(scope="module")
def text():
    return _fastq(np.random.default_rng(1), 60000)  # ~19 MB of FASTQ


def _roundtrip(tmp_path, name, blob, want, monkeypatch, chunk=None, threads=0):
    p = tmp_path / name
    p.write_bytes(blob)
    if chunk:
        _hip.debug_set("pgzip_chunk", chunk)
    got = _hip.gunzip_file(str(p), nthreads=threads)
    assert len(got) == len(want) and got == want, name


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("chunk", [64 << 10, 300_000, 2 << 20])
def test_one_member_at_every_level_and_chunk_size(tmp_path, text, monkeypatch, level, chunk):
    _roundtrip(tmp_path, "a.gz", gzip.compress(text, level), text, monkeypatch, chunk=chunk, threads=6)


def test_members_padding_and_odd_shapes(tmp_path, text, monkeypatch):
    a, b, c = gzip.compress(text[:5_000_000], 6), gzip.compress(text[5_000_000:5_000_100], 1), gzip.compress(text[5_000_100:], 4)
    _roundtrip(tmp_path, "members.gz", a + b + c, text, monkeypatch, chunk=100_000, threads=5)
    _roundtrip(tmp_path, "padded.gz", a + b + c + b"\0" * 4000, text, monkeypatch, chunk=100_000, threads=5)
    for pad in (b"\n", b"\0" * 3, b"\0" * 9, b"x" * 9):  # (fewer bytes than a member header has: still garbage, not a truncated member)
        _roundtrip(tmp_path, "padded%d.gz" % len(pad), a + b + c + pad, text, monkeypatch, chunk=100_000, threads=5)
    _roundtrip(tmp_path, "garbage.gz", a + b"trailing garbage", text[:5_000_000], monkeypatch, chunk=100_000, threads=3)
    _roundtrip(tmp_path, "empty_member.gz", gzip.compress(b"") + a + gzip.compress(b""), text[:5_000_000], monkeypatch, chunk=64 << 10)
    _roundtrip(tmp_path, "tiny.gz", gzip.compress(b"@r\nACGT\n+\nIIII\n"), b"@r\nACGT\n+\nIIII\n", monkeypatch)
    _roundtrip(tmp_path, "nothing.gz", gzip.compress(b""), b"", monkeypatch)
    # header fields: file name, comment, extra, header CRC
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(text[:3_000_000]) + co.flush()
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16]) + b"\0\0\0\0\0\x03" + b"\x05\0hello" + b"name.fq\0" + b"a comment\0"
    trailer = zlib.crc32(text[:3_000_000]).to_bytes(4, "little") + (3_000_000).to_bytes(4, "little")
    _roundtrip(tmp_path, "fields.gz", hdr + raw + trailer, text[:3_000_000], monkeypatch, chunk=80_000, threads=4)
    # stored blocks (level 0) and fixed-Huffman blocks (tiny inputs): no dynamic block to enter at — still right
    _roundtrip(tmp_path, "stored.gz", gzip.compress(text[:2_000_000], 0), text[:2_000_000], monkeypatch, chunk=64 << 10, threads=4)
    # one thread
    _roundtrip(tmp_path, "one_thread.gz", gzip.compress(text, 6), text, monkeypatch, chunk=200_000, threads=1)


def test_binary_data_is_still_inflated_correctly(tmp_path, monkeypatch):
    """Not text: no block start passes the trial decode (literals must be text), so the first thread runs through all of it."""
    rng = np.random.default_rng(3)
    blob = (rng.integers(0, 256, size=400_000, dtype=np.uint8).tobytes() + bytes(300_000)) * 3
    _roundtrip(tmp_path, "bin.gz", gzip.compress(blob, 6), blob, monkeypatch, chunk=64 << 10, threads=4)
    # text with a little binary in the middle
    t = _fastq(rng, 8000)
    mixed = t + bytes(range(256)) * 50 + t
    _roundtrip(tmp_path, "mixed.gz", gzip.compress(mixed, 6), mixed, monkeypatch, chunk=64 << 10, threads=4)


def test_corrupt_and_truncated_streams_are_errors(tmp_path, text, monkeypatch):
    _hip.debug_set("pgzip_chunk", 100_000)
    blob = gzip.compress(text[:6_000_000], 6)
    for name, bad in (("cut.gz", blob[:-9]), ("cut_mid.gz", blob[: len(blob) // 2]), ("cut_header.gz", blob[:6]),
                      ("crc.gz", blob[:-8] + b"\0\0\0\0" + blob[-4:]), ("isize.gz", blob[:-4] + b"\1\0\0\0")):
        p = tmp_path / name
        p.write_bytes(bad)
        with pytest.raises(OSError):
            _hip.gunzip_file(str(p), nthreads=4)
    flipped = bytearray(blob)
    flipped[len(blob) // 3] ^= 0x55  # a flipped byte mid-stream: caught by the decoder or by the CRC
    p = tmp_path / "flip.gz"
    p.write_bytes(bytes(flipped))
    with pytest.raises(OSError):
        _hip.gunzip_file(str(p), nthreads=4)
    p = tmp_path / "notgz"
    p.write_bytes(b"plain text, not gzip\n" * 100)
    with pytest.raises(OSError):
        _hip.gunzip_file(str(p))
    with pytest.raises(OSError):
        _hip.gunzip_file(str(tmp_path / "missing.gz"))


def test_rate_against_zlib(tmp_path, monkeypatch):
    """127 MB through both, rates printed (pytest -s).  No pass / fail on speed: the CPU suite's container is an 8-core microVM
    in which the first touch of fresh memory costs more than the inflating (the same call takes 0.26 s in a small process and
    2 s in one that has already grown; tools/gzip_probe.py measures the real thing on the GPU box's host)."""
    import time
    text = _fastq(np.random.default_rng(9), 400_000)  # ~127 MB
    p = tmp_path / "big.gz"
    p.write_bytes(gzip.compress(text, 6))
    t0 = time.perf_counter()
    want = zlib.decompress(p.read_bytes(), 47)
    t_z = time.perf_counter() - t0
    t0 = time.perf_counter()
    got = _hip.gunzip_file(str(p))
    t_p = time.perf_counter() - t0
    assert got == want == text
    print("zlib %.2f s (%.0f MB/s), parallel %.2f s (%.0f MB/s) on %d cores" % (t_z, len(text) / t_z / 1e6, t_p, len(text) / t_p / 1e6, os.cpu_count()))
