"""CPU: what the streaming readers hand to the device for a plain FASTQ / SAM file (mg_stream_thin_file, the host code of
mg_sketch_stream_add_file / mg_sam_stream_file) against a Python restatement of the rule — a FASTQ record becomes ">" and
its sequence line, a SAM line with at least 11 fields of str.split() keeps every byte but its SEQ (mark + length) and QUAL
('*') — for every way a piece can cut a line, and the errors the device parser raises for the same text."""
import os
import re

import numpy as np
import pytest

from metalign_amd import _hip

MARK = b"\x01"
WS = b" \t\n\r\x0b\x0c\x1c\x1d\x1e\x1f"


def thin_fastq(text):
    lines = text.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()  # (nothing behind the last newline is not a line)
    nrec = len(lines) // 4
    for ln in lines[4 * nrec:]:
        assert ln in (b"", b"\r"), "leftover"
    out = []
    for r in range(nrec):
        h, s, p = lines[4 * r], lines[4 * r + 1], lines[4 * r + 2]
        assert h.rstrip(b"\r")[:1] == b"@" and p.rstrip(b"\r")[:1] == b"+", "malformed"
        out.append(b">\n" + s + b"\n")
    return b"".join(out)


def thin_sam_line(line):
    if not line or line[:1] == b"@":
        return line
    spans = [(m.start(), m.end()) for m in re.finditer(rb"[^ \t\n\r\x0b\x0c\x1c-\x1f]+", line)]
    if len(spans) < 11:
        return line
    (b9, e9), (b10, e10) = spans[9], spans[10]
    seq = b"*" if line[b9:e9] == b"*" else MARK + str(e9 - b9).encode()
    return line[:b9] + seq + line[e9:b10] + b"*" + line[e10:]


def thin_sam(text):
    parts = text.split(b"\n")
    tail = parts.pop()  # what follows the last newline (a last line without one), or b""
    out = b"".join(thin_sam_line(ln) + b"\n" for ln in parts)
    return out + (thin_sam_line(tail) if tail else b"")


def run(tmp_path, kind, text, piece, threads=3):
    src, dst = tmp_path / ("in." + kind), tmp_path / ("out." + kind)
    src.write_bytes(text)
    _hip.thin_file(str(src), kind, str(dst), piece_bytes=piece, nthreads=threads)
    return dst.read_bytes()


def _fastq(rng, n, crlf=False, maxlen=300):
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(n):
        L = int(rng.integers(0, maxlen))
        seq = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), size=L))
        qual = bytes(rng.integers(33, 74, size=L, dtype=np.uint8))  # may begin with '@' or '+', as real quality lines do
        out.append(b"@r%d some text" % i + nl + seq + nl + b"+" + (b"r%d" % i if i % 3 == 0 else b"") + nl + qual + nl)
    return b"".join(out)


@pytest.mark.parametrize("crlf", [False, True])
def test_fastq_pieces_cut_anywhere(tmp_path, crlf):
    rng = np.random.default_rng(4)
    text = _fastq(rng, 1500, crlf)
    want = thin_fastq(text)
    assert want.count(b">\n") == 1500
    for piece in (1 << 16, (1 << 16) + 4096, 3 << 16, 1 << 20):
        for threads in (1, 5):
            assert run(tmp_path, "fastq", text, piece, threads) == want, (piece, threads)
    # the file's end: blank lines after the last record, a last line without its newline
    assert run(tmp_path, "fastq", text + b"\n\r\n", 1 << 16) == want
    assert run(tmp_path, "fastq", text[:-1], 1 << 16) == thin_fastq(text[:-1]) == want
    assert run(tmp_path, "fastq", b"", 1 << 16) == b""
    assert run(tmp_path, "fastq", b"\n\n", 1 << 16) == b""


def test_fastq_refused_like_the_device_parser(tmp_path):
    rng = np.random.default_rng(5)
    text = _fastq(rng, 900)
    bad_sep = text.replace(b"\n+\n", b"\n-\n", 500).replace(b"\n-\n", b"\n+\n", 499)
    with pytest.raises(_hip.HipError) as e:
        run(tmp_path, "fastq", bad_sep, 1 << 16)
    assert e.value.code == _hip.ERR_ARG and "malformed record" in str(e.value)
    cut = text[: text.rindex(b"\n+")]  # a record without its last two lines
    with pytest.raises(_hip.HipError) as e:
        run(tmp_path, "fastq", cut, 1 << 16)
    assert e.value.code == _hip.ERR_ARG and "whole number" in str(e.value)
    with pytest.raises(_hip.HipError) as e:
        run(tmp_path, "fastq", text.replace(b"@r700 ", b"r700 "), 1 << 16)
    assert "malformed record 700" in str(e.value)
    # a record longer than what a piece leaves room for: the stream's capacity error (select_db then reads piece-wise)
    long = b"@a\n" + b"ACGT" * 30000 + b"\n+\n" + b"I" * 120000 + b"\n" + text
    with pytest.raises(_hip.HipError) as e:
        run(tmp_path, "fastq", long, 1 << 16)
    assert e.value.code == _hip.ERR_CAPACITY
    assert run(tmp_path, "fastq", long, 1 << 20) == thin_fastq(long)


def _sam(rng, n, sep=b"\t"):
    lines = [b"@HD\tVN:1.6\tSO:unsorted", b"@SQ\tSN:ref1\tLN:5000"]
    for i in range(n):
        L = int(rng.choice([0, 1, 2, 9, 10, 150, 151, 1000]))
        seq = b"*" if L == 0 else bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L))
        qual = b"*" if L == 0 or i % 7 == 0 else bytes(rng.integers(33, 74, size=L, dtype=np.uint8))
        fields = [b"read%d" % (i // 2), str(int(rng.choice([0, 16, 256, 4, 99, 147]))).encode(), b"ref%d" % (i % 5), b"%d" % (i + 1), b"60",
                  b"%dM" % max(L, 1), b"*", b"0", b"0", seq, qual, b"NM:i:%d" % (i % 4), b"AS:i:77"]
        k = int(rng.choice([13, 13, 13, 12, 11, 10, 5, 1]))
        lines.append(sep.join(fields[:k]))
    lines.insert(40, b"")
    lines.insert(90, b"   ")
    return b"\n".join(lines) + b"\n"


@pytest.mark.parametrize("sep", [b"\t", b" ", b" \t ", b"\x1c"])
def test_sam_pieces_cut_anywhere(tmp_path, sep):
    rng = np.random.default_rng(6)
    text = _sam(rng, 1200, sep)
    want = thin_sam(text)
    assert len(want) < 0.6 * len(text) and MARK in want
    for piece in (1 << 16, (1 << 16) + 8192, 1 << 20):
        for threads in (1, 4):
            assert run(tmp_path, "sam", text, piece, threads) == want, (piece, threads)
    assert run(tmp_path, "sam", text[:-1], 1 << 16) == thin_sam(text[:-1])  # the last line without its newline
    assert run(tmp_path, "sam", text + b"\r\n\n", 1 << 16) == want + b"\r\n\n"
    # the fields the tokeniser reads are what they were: same str.split() but for SEQ and QUAL
    for a, b in zip(text.split(b"\n"), want.split(b"\n")):
        fa, fb = a.decode("latin-1").split(), b.decode("latin-1").split()  # (str.split: \x1c-\x1f are white space, as in the reference)
        fa, fb = [x.encode("latin-1") for x in fa], [x.encode("latin-1") for x in fb]
        assert len(fa) == len(fb)
        for j, (x, y) in enumerate(zip(fa, fb)):
            if len(fa) >= 11 and a[:1] != b"@" and j == 9:
                assert y == (b"*" if x == b"*" else MARK + str(len(x)).encode())
            elif len(fa) >= 11 and a[:1] != b"@" and j == 10:
                assert y == b"*"
            else:
                assert x == y


def test_sam_lines_that_grow_and_lines_longer_than_a_piece(tmp_path):
    one = b"r\t0\tref\t1\t60\t1M\t*\t0\t0\tA\tI\tNM:i:0\n"  # SEQ and QUAL of one character: the thinned line is a byte longer
    assert len(thin_sam(one)) == len(one) + 1
    text = one * 50 + b"r\t0\tref\t1\t60\t150M\t*\t0\t0\t" + b"A" * 150 + b"\t" + b"I" * 150 + b"\tNM:i:0\n"
    assert run(tmp_path, "sam", text, 1 << 16) == thin_sam(text)
    with pytest.raises(_hip.HipError) as e:  # more growth than the room in front of a piece
        run(tmp_path, "sam", one * 12000, 1 << 18)  # (a piece of 6800 such lines; 4096 bytes of room)
    assert "grew" in str(e.value)
    assert run(tmp_path, "sam", one * 12000, 1 << 16) == thin_sam(one * 12000)  # (1600 of them per piece: fits)
    long_line = b"r\t0\tref\t1\t60\t90000M\t*\t0\t0\t" + b"A" * 90000 + b"\t" + b"I" * 90000 + b"\tNM:i:0\n"
    text = one * 10 + long_line + one * 10
    with pytest.raises(_hip.HipError) as e:
        run(tmp_path, "sam", text, 1 << 16)
    assert e.value.code == _hip.ERR_CAPACITY
    assert run(tmp_path, "sam", text, 1 << 20) == thin_sam(text)
    # a line that runs through whole pieces: the pieces in between own no line start
    assert run(tmp_path, "sam", b"@CO\t" + b"x" * 400000 + b"\n" + one, 1 << 20) == b"@CO\t" + b"x" * 400000 + b"\n" + thin_sam(one)


def test_the_byte_by_byte_scanner_gives_the_same(tmp_path):
    """The scanners take 32 bytes per step where the host has AVX2; the knob no_avx2 (read once per process) selects the plain
    loop: this module again in a process of its own."""
    import subprocess
    import sys
    if "no_avx2=1" in os.environ.get("MG_TEST_KNOBS", ""):
        pytest.skip("already the plain loop")
    env = dict(os.environ, MG_TEST_KNOBS="no_avx2=1")  # (tests/conftest.py hands it to mg_debug_set when the child's session starts)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", os.path.abspath(__file__), "-p", "no:cacheprovider",
                        "-k", "not byte_by_byte"], env=env, capture_output=True, text=True, timeout=600,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
