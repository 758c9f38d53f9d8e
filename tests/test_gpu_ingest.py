"""-m gpu: device-side ingest (FASTQ / FASTA -> bases + offsets, SAM text -> records) against the host parsers,
which the CPU suite pins to the reference's behaviour."""
import numpy as np
import pytest

import samgen
from metalign_amd import _hip, formats
from metalign_amd import map_and_profile as mp

pytestmark = pytest.mark.gpu


def _fastq_text(rng, n, crlf=False, trailing=True):
    nl = "\r\n" if crlf else "\n"
    out = []
    for i in range(n):
        L = int(rng.integers(0, 260)) if i % 17 else 0
        seq = "".join(rng.choice(list("ACGTNacgt"), size=L))
        out.append("@read%d some description%s%s%s+%s%s%s" % (i, nl, seq, nl, nl, "I" * L, nl))
    text = "".join(out)
    return text if trailing else text.rstrip("\r\n")


@pytest.mark.parametrize("crlf,trailing,extra_blank", [(False, True, False), (True, True, False), (False, False, False),
                                                       (False, True, True)])
def test_fastq_parse_matches_host_reader(hip, tmp_path, crlf, trailing, extra_blank):
    rng = np.random.default_rng(4)
    text = _fastq_text(rng, 5000, crlf, trailing) + ("\n" if extra_blank else "")
    p = tmp_path / "r.fq"
    p.write_bytes(text.encode())
    want_b, want_o, _ = formats.read_sequences(str(p), "fastq")
    rd = hip.parse_reads(text.encode(), "fastq")
    b, o = rd.download()
    assert rd.count == len(want_o) - 1
    assert np.array_equal(o, want_o) and np.array_equal(b, want_b)


def test_fasta_single_line_and_multiline_fallback(hip, tmp_path):
    from metalign_amd import select_db
    rng = np.random.default_rng(5)
    seqs = ["".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 400)))) for _ in range(2000)]
    one = "".join(">s%d x\n%s\n" % (i, s) for i, s in enumerate(seqs))
    p1 = tmp_path / "one.fa"
    p1.write_text(one)
    wb, wo, _ = formats.read_sequences(str(p1), "fasta")
    rd = select_db.load_reads_device(hip, str(p1), "fasta")
    assert isinstance(rd, _hip.Reads)
    b, o = rd.download()
    assert np.array_equal(o, wo) and np.array_equal(b, wb)
    multi = "".join(">s%d\n%s\n" % (i, "\n".join(s[j:j + 60] for j in range(0, len(s), 60))) for i, s in enumerate(seqs))
    p2 = tmp_path / "multi.fa"
    p2.write_text(multi)
    rd2 = select_db.load_reads_device(hip, str(p2), "fasta")  # host fallback, same content
    d_b, d_o = rd2.device_ptrs()
    assert rd2.count == len(seqs)
    hs = hip.sketch_reads_dev(d_b, d_o, rd2.count, 21, _hip.U64_MAX, 0)
    h1 = hs.download()
    d_b1, d_o1 = rd.device_ptrs()
    hs1 = hip.sketch_reads_dev(d_b1, d_o1, rd.count, 21, _hip.U64_MAX, 0)
    assert np.array_equal(h1[0], hs1.download()[0])
    with pytest.raises(_hip.HipError):
        hip.parse_reads(b"@r1\nACGT\nIIII\n", "fastq")          # 3 lines: not whole records
    with pytest.raises(_hip.HipError):
        hip.parse_reads(b"@r1\nACGT\n-\nIIII\n", "fastq")       # bad separator line


def _acc_index(accs):
    idx = {"Unmapped": 0}
    idx.update({a: i + 1 for i, a in enumerate(accs)})
    return idx


@pytest.mark.parametrize("kind,seed,n", [("single", 3, 20000), ("paired", 4, 12000)])
def test_sam_tokeniser_matches_host(hip, monkeypatch, kind, seed, n):
    dbtext, accs, taxids = samgen.make_dbinfo(seed=8, n_species=25)
    gen = samgen.make_sam_single if kind == "single" else samgen.make_sam_paired
    text = gen(seed, n, accs, taxids)
    # whitespace variants the reference's strip().split() accepts
    lines = text.splitlines(True)
    lines[10] = "  " + lines[10].replace("\t", "  ", 3)
    lines[11] = lines[11].rstrip("\n") + " \t\r\n"
    lines.insert(12, "\n")
    lines.insert(13, "q\t0\tx\n")
    lines.insert(14, "@CO\tcomment with enough\tfields\ta\tb\tc\td\te\tf\tg\th\ti\n")
    text = "".join(lines)
    idx = _acc_index(accs)
    want = mp.tokenise_sam(text.splitlines(True), idx)
    got = mp.tokenise_sam_device(iter(text.splitlines(True)), idx)  # iterator path, one chunk
    assert np.array_equal(got, want)
    monkeypatch.setattr(mp, "_CHUNK_BYTES", 40000)                 # many chunks: QNAME carried across
    import io
    got2 = mp.tokenise_sam_device(io.BytesIO(text.encode()), idx)
    assert np.array_equal(got2, want)
    got3 = mp.tokenise_sam_device(iter(text.splitlines(True)), idx)
    assert np.array_equal(got3, want)


@pytest.mark.parametrize("kind,seed,n", [("single", 5, 6000), ("paired", 6, 4000)])
def test_sam_pieces_tokenised_on_the_device_give_the_whole_files_records(hip, tmp_path, kind, seed, n):
    """What the ranks of a multi-GPU map_and_profile do, on one GPU: line-aligned byte ranges of the SAM file go up
    (Hip.upload_file with offset / length) and are tokenised independently; with the new-read bit cleared wherever a
    piece's first retained QNAME equals the last retained QNAME in front of it, the concatenation is the record stream
    of the whole file (what rank 0 then runs stage C on) — for several world sizes, with multi-line reads across cuts."""
    dbtext, accs, taxids = samgen.make_dbinfo(seed=8, n_species=25)
    text = (samgen.make_sam_single if kind == "single" else samgen.make_sam_paired)(seed, n, accs, taxids)
    path = tmp_path / "x.sam"
    path.write_text(text)
    idx = _acc_index(accs)
    want = mp.tokenise_sam(text.splitlines(True), idx)
    names = [None] * len(idx)
    for a, i in idx.items():
        names[i] = a
    index = hip.acc_index(names)
    try:
        for world in (1, 2, 3, 7, 40):
            pieces, firsts, lasts = [], [], []
            for r in range(world):
                a, b = mp.sam_range_of_rank(str(path), r, world)
                d_text, size = hip.upload_file(str(path), offset=a, length=b - a)
                batch = hip.sam_tokenize_dev_batch(d_text.ptr, size, index, "")
                recs = np.zeros(batch.count, dtype=mp._hip.REC_DTYPE)
                if batch.count:
                    hip._chk(hip.lib.mg_sam_batch_download(batch.handle, recs.ctypes.data_as(__import__("ctypes").c_void_p)))
                pieces.append(recs)
                firsts.append((mp.first_retained_qname(str(path), a, b) or "") if batch.count else "")
                lasts.append(batch.last_qname if batch.count else "")
                batch.free()
                d_text.free()
            got = np.concatenate(pieces)

            def clear(i):
                got["ref_new"][i] &= 0x7FFFFFFF
            mp.clear_continued_heads(clear, [len(p) for p in pieces], firsts, lasts)
            assert np.array_equal(got, want), world
    finally:
        index.free()


def test_sam_tokeniser_raises_what_the_reference_raises(hip):
    dbtext, accs, taxids = samgen.make_dbinfo(seed=8, n_species=5)
    idx = _acc_index(accs)
    ok = samgen._line("r1", 0, accs[0], "40M", "ACGT" * 10, 0)
    cases = [
        (samgen._line("r2", 0, "NOPE.1", "40M", "ACGT" * 10, 0), KeyError),
        ("r3\t0\t%s\t1\t60\t40M\t*\t0\t0\tACGT\n" % accs[1], IndexError),
        (samgen._line("r4", 0, accs[0], "20=20X", "ACGT" * 10, 0), ValueError),
        (samgen._line("r5", "zz", accs[0], "40M", "ACGT" * 10, 0), ValueError),
        ("r6\t0\t%s\t1\t60\t40\t*\t0\t0\tACGT\tIIII\tNM:i:0\n" % accs[1], ZeroDivisionError),
        ("r7\t0\t%s\t1\t60\t40M\t*\t0\t0\tACGT\tIIII\tNM:i\n" % accs[1], ValueError),
    ]
    for bad, exc in cases:
        text = ok * 3 + bad + ok
        with pytest.raises(exc) as e1:
            mp.tokenise_sam(text.splitlines(True), idx)
        with pytest.raises(exc) as e2:
            mp.tokenise_sam_device(iter(text.splitlines(True)), idx)
        assert str(e1.value) == str(e2.value)
    # unmapped / header / short lines are skipped before any field is parsed
    text = "@HD\tx\n" + "u\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\n" + "a b c\n" + ok
    assert np.array_equal(mp.tokenise_sam_device(iter(text.splitlines(True)), idx), mp.tokenise_sam(text.splitlines(True), idx))


@pytest.mark.gpu
def test_upload_file_chunks(hip, tmp_path):
    """Hip.upload_file: a file through page-locked chunks filled by reader threads (several rounds of two buffers, of
    four above eight chunks, a ragged tail, the empty file, chunks kept from one call to the next) == its bytes."""
    rng = np.random.default_rng(3)
    for size, chunk in ((0, 1 << 16), (1, 1 << 16), (300_001, 1 << 16), (1 << 16, 1 << 16), (5 * (1 << 16) + 7, 1 << 16),
                        (19 * (1 << 16) + 123, 1 << 16), (3 * (1 << 16), 1 << 16), (40_000, 1 << 16)):
        data = rng.integers(0, 256, size=size, dtype=np.uint8)
        path = tmp_path / ("f%d.bin" % size)
        path.write_bytes(data.tobytes())
        dev, n = hip.upload_file(str(path), chunk=chunk)
        assert n == size
        if size:
            assert np.array_equal(dev.download()[:size], data)
        dev.free()


@pytest.mark.gpu
def test_multiline_fasta_on_device_equals_host_parser(hip, tmp_path):
    """mg_reads_parse_dev format 2 == formats.read_sequences on FASTA with wrapped sequences, CRLF, blank and indented
    lines, an empty record, junk in front of the first header and no newline at the end."""
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    for trial, (nl, tail) in enumerate((("\n", "\n"), ("\r\n", ""), ("\n", ""))):
        lines = ["junk before any header", ""] if trial != 2 else []
        for r in range(60):
            lines.append(">seq%d some description" % r)
            n = 0 if r == 7 else int(rng.integers(1, 400))
            seq = alphabet[rng.integers(0, len(alphabet), size=n)].tobytes().decode()
            width = int(rng.integers(1, 80))
            for i in range(0, n, width):
                pad = "  " if rng.random() < 0.1 else ""
                lines.append(pad + seq[i:i + width] + ("\t" if rng.random() < 0.1 else ""))
            if rng.random() < 0.2:
                lines.append("")
        text = nl.join(lines) + tail
        path = tmp_path / ("ml%d.fa" % trial)
        path.write_bytes(text.encode())
        want_b, want_o, _ = formats.read_sequences(str(path), "fasta")
        reads = hip.parse_reads(text.encode(), "fasta_ml")
        got_b, got_o = reads.download()
        assert reads.count == len(want_o) - 1
        assert np.array_equal(got_o, want_o) and np.array_equal(got_b[: int(want_o[-1])], want_b)
        reads.free()
    empty = hip.parse_reads(b"", "fasta_ml")
    assert empty.count == 0
    empty.free()
