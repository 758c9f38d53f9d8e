"""GPU: gzip / BGZF inflated on the device (mg_inflate.hip, through the C ABI) against Python's zlib: the corpus of
tests/test_pgzip.py (levels, members, padding, header fields, stored blocks, binary data, truncated / corrupt streams refused),
BGZF, many stages, jobs that overflow their reservation, and the streaming entry points on `.gz` files."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fastq(rng, nreads, readlen=150):
    seqs = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(nreads, readlen))
    qual = rng.integers(35, 74, size=(nreads, readlen)).astype(np.uint8)
    out = []
    for i in range(nreads):
        out.append(b"@read%d/1 len=%d\n" % (i, readlen) + seqs[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n")
    return b"".join(out)


def bgzf(data, block=65280, level=6, eof=True):
    out = []
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof else [])
    for c in chunks:
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = co.compress(c) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(raw) + 8 - 1) + raw + struct.pack("<II", zlib.crc32(c), len(c)))
    return b"".join(out)


@pytest.fixture(scope="module")
def text():
    return _fastq(np.random.default_rng(1), 60000)  # ~19 MB of FASTQ


@pytest.fixture(params=["job per wavefront", "job per lane"])
def cfg(hip, request):
    """Both decoders on every case: launches of any size decode a job per lane (lane_jobs = 0) or a job per wavefront (never per lane)."""
    hip.inflate_config(lane_jobs=0 if request.param == "job per lane" else 1 << 40)
    yield hip
    hip.inflate_config(chunk_bytes=32 << 10, stage_bytes=-1, ratio=10, on=1, lane_jobs=1 << 40)


def _same(hip, blob, want, what):
    got = hip.inflate(blob)
    assert len(got) == len(want), "%s: %d bytes against %d" % (what, len(got), len(want))
    if got != want:
        a, b = np.frombuffer(got, np.uint8), np.frombuffer(want, np.uint8)
        bad = np.flatnonzero(a != b)
        raise AssertionError("%s: %d bytes differ, first at %d (got %r, want %r)" % (what, bad.size, bad[0], got[bad[0] - 20: bad[0] + 20], want[bad[0] - 20: bad[0] + 20]))


@pytest.mark.parametrize("level", [1, 6, 9])
@pytest.mark.parametrize("chunk", [8 << 10, 32 << 10, 300_000])
def test_one_member_at_every_level_and_chunk_size(cfg, text, level, chunk):
    cfg.inflate_config(chunk_bytes=chunk)
    cfg.inflate_stats(reset=True)
    _same(cfg, gzip.compress(text, level), text, "level %d, chunks of %d" % (level, chunk))
    st = cfg.inflate_stats()
    assert st["stages"] == 1 and (st["jobs"] > 10 or chunk > 100_000), st  # the stream was entered in the middle


def test_many_stages_and_jobs_that_overflow_their_reservation(cfg, text):
    blob = gzip.compress(text, 6)
    cfg.inflate_config(chunk_bytes=8 << 10, stage_bytes=300_000)
    cfg.inflate_stats(reset=True)
    _same(cfg, blob, text, "stages of 300 kB")
    assert cfg.inflate_stats()["stages"] >= 10
    cfg.inflate_config(chunk_bytes=16 << 10, stage_bytes=1 << 20, ratio=1)  # nothing inflates 1:1: every job counts on and is decoded again
    cfg.inflate_stats(reset=True)
    _same(cfg, blob, text, "ratio 2")
    assert cfg.inflate_stats()["redone"] > 3
    zeros = bytes(30 << 20)  # 1000:1
    cfg.inflate_config(chunk_bytes=8 << 10, stage_bytes=128 << 20, ratio=10)
    _same(cfg, gzip.compress(zeros + text[:1_000_000] + zeros, 6), zeros + text[:1_000_000] + zeros, "zeros")


def test_members_padding_and_odd_shapes(cfg, text):
    cfg.inflate_config(chunk_bytes=16 << 10, stage_bytes=2 << 20)
    a, b, c = gzip.compress(text[:5_000_000], 6), gzip.compress(text[5_000_000:5_000_100], 1), gzip.compress(text[5_000_100:], 4)
    _same(cfg, a + b + c, text, "members")
    for pad in (b"\0", b"\0" * 9, b"\0" * 4000, b"\n"):
        _same(cfg, a + b + c + pad, text, "padding of %d" % len(pad))
    _same(cfg, a + b"trailing garbage", text[:5_000_000], "garbage")
    _same(cfg, gzip.compress(b"") + a + gzip.compress(b""), text[:5_000_000], "empty members")
    _same(cfg, gzip.compress(b"@r\nACGT\n+\nIIII\n"), b"@r\nACGT\n+\nIIII\n", "tiny")
    _same(cfg, gzip.compress(b""), b"", "nothing")
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = co.compress(text[:3_000_000]) + co.flush()
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16]) + b"\0\0\0\0\0\x03" + b"\x05\0hello" + b"name.fq\0" + b"a comment\0"
    trailer = zlib.crc32(text[:3_000_000]).to_bytes(4, "little") + (3_000_000).to_bytes(4, "little")
    _same(cfg, hdr + raw + trailer, text[:3_000_000], "header fields")
    _same(cfg, gzip.compress(text[:2_000_000], 0), text[:2_000_000], "stored blocks")
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_FIXED)
    _same(cfg, co.compress(text[:1_000_000]) + co.flush(), text[:1_000_000], "fixed blocks")
    many = b"".join(gzip.compress(text[i:i + 3000], 6) for i in range(0, 1_500_000, 3000))  # 500 members, not BGZF
    _same(cfg, many, text[:1_500_000], "500 members")


def test_binary_data(cfg):
    rng = np.random.default_rng(3)
    blob = (rng.integers(0, 256, size=400_000, dtype=np.uint8).tobytes() + bytes(300_000)) * 3
    cfg.inflate_config(chunk_bytes=16 << 10)
    _same(cfg, gzip.compress(blob, 6), blob, "binary")
    t = _fastq(rng, 8000)
    mixed = t + bytes(range(256)) * 50 + t
    _same(cfg, gzip.compress(mixed, 6), mixed, "mixed")


def test_bgzf(cfg, text):
    cfg.inflate_stats(reset=True)
    _same(cfg, bgzf(text), text, "bgzf")
    assert cfg.inflate_stats()["jobs"] >= len(text) // 65280
    _same(cfg, bgzf(text, eof=False), text, "bgzf without the EOF block")
    _same(cfg, bgzf(text[:100_000], block=1000, level=1), text[:100_000], "bgzf, small blocks")
    _same(cfg, bgzf(b""), b"", "empty bgzf")
    bad = bytearray(bgzf(text[:400_000]))
    bad[70_000] ^= 0x10
    with pytest.raises(OSError):
        cfg.inflate(bytes(bad))


def test_corrupt_and_truncated_streams_are_errors(cfg, text):
    cfg.inflate_config(chunk_bytes=16 << 10, stage_bytes=1 << 20)
    blob = gzip.compress(text[:6_000_000], 6)
    for name, bad in (("cut", blob[:-9]), ("cut_mid", blob[: len(blob) // 2]), ("cut_header", blob[:6]),
                      ("crc", blob[:-8] + b"\0\0\0\0" + blob[-4:]), ("isize", blob[:-4] + b"\1\0\0\0"), ("not gzip", b"plain text, not gzip\n" * 100),
                      ("empty", b"")):
        with pytest.raises(OSError):
            cfg.inflate(bad)
    rng = np.random.default_rng(5)
    for _ in range(6):
        flipped = bytearray(blob)
        flipped[int(rng.integers(100, len(blob) - 100))] ^= 1 << int(rng.integers(0, 8))
        with pytest.raises(OSError):
            cfg.inflate(bytes(flipped))
    _same(cfg, blob, text[:6_000_000], "the library is in order after the errors")


def test_streamed_gz_files_give_what_the_plain_files_give(cfg, tmp_path):
    """mg_sketch_stream_add_file on .gz (gzip, many members, BGZF) = the reads in one launch; many small stages (the unfinished record of
    a stage is carried in front of the next stage's text on the device) and one; the device inflater on and off."""
    import test_gpu_stream as tgs
    hip = cfg
    ks = [21, 51]
    gb, go, rb, ro = tgs._sample(2, nreads=20000)
    tabs, hmaxs, filts = tgs._tables(hip, gb, go, ks)
    want = tgs._whole(hip, rb, ro, ks, hmaxs, filts)
    fq = tgs._fastq(rb, ro)
    forms = {"one.fq.gz": gzip.compress(fq, 6), "bgzf.fq.gz": bgzf(fq), "padded.fq.gz": gzip.compress(fq, 1) + b"\0" * 1000,
             "members.fq.gz": b"".join(gzip.compress(fq[a: a + 700001], 1) for a in range(0, len(fq), 700001))}
    for name, data in forms.items():
        p = tmp_path / name
        p.write_bytes(data)
        for stage, on in ((100_000, 1), (128 << 20, 1), (0, 0)):
            hip.inflate_config(chunk_bytes=16 << 10, stage_bytes=stage, on=on)
            hip.inflate_stats(reset=True)
            got, counts = tgs._streamed(hip, ks, hmaxs, filts, rb.size, lambda st: st.add_file(str(p), "fastq"))
            assert counts == (len(ro) - 1, rb.size), (name, stage, on)
            tgs._same(got, want)
            st = hip.inflate_stats()
            assert (st["jobs"] > 0) == (on == 1) and (stage != 100_000 or name.startswith("bgzf") or st["stages"] > 5), (name, stage, on, st)
        # a file larger than the device inflater may hold (the knob stands in for "half of the free device memory"): the host
        # inflater takes it, same reads
        from metalign_amd import _hip as _h
        hip.inflate_config(chunk_bytes=16 << 10, stage_bytes=128 << 20, on=1)
        _h.debug_set("inflate_dev_max_bytes", 1000)
        try:
            hip.inflate_stats(reset=True)
            got, counts = tgs._streamed(hip, ks, hmaxs, filts, rb.size, lambda st: st.add_file(str(p), "fastq"))
            assert counts == (len(ro) - 1, rb.size) and hip.inflate_stats()["jobs"] == 0, name
            tgs._same(got, want)
        finally:
            _h.debug_set("inflate_dev_max_bytes", 0)


def test_random_streams_and_damaged_ones_against_zlib(hip):
    """tools/inflate_soak.py in small: random payloads (text, FASTQ-like, noise, runs), levels, members, flushes, Z_FIXED / Z_HUFFMAN_ONLY /
    Z_RLE, random chunk / stage / ratio settings and both decoders — equal to what zlib compressed; and the same damaged (bits flipped,
    cut): refused where zlib refuses, read alike where it does not."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("inflate_soak", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "inflate_soak.py"))
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    sizes = ([0, 1, 100, 5000, 70_000, 400_000], [.05, .05, .2, .3, .25, .15])
    try:
        total, _ = soak.soak(hip, 60, 314, False, sizes)
        assert total > 1_000_000
        _, refused = soak.soak(hip, 60, 315, True, sizes)
        assert refused > 30
    finally:
        hip.inflate_config(chunk_bytes=32 << 10, stage_bytes=-1, ratio=10, on=1, lane_jobs=1 << 40)
