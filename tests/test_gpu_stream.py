"""-m gpu: the streamed read sketch (mg_sketch_stream_*, metalign_amd/csrc/mg_stream.hip).

A sample that arrives in pieces — a reads file (plain, gzip, BGZF) read by reader threads into page-locked chunks, chunk
i + 1 uploaded while chunk i is parsed on the device and hashed into ONE set of counting tables — must give, bit for bit,
the sketch of the whole sample: against the one-shot path (mg_sketch_reads_multi_dev_async on the whole parsed file) and
against the oracle.  Replaces kmc reading the reads file, scripts/select_db.py:45-52 (`.gz`: :146-148)."""
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

import util
from metalign_amd import _hip

pytestmark = pytest.mark.gpu


def _sample(seed, nreads=30000, readlen=150, ragged=False, with_n=False):
    rng = np.random.default_rng(seed)
    gb, go = util.random_genomes(rng, 40, 6000)
    rb, ro, _ = util.sample_reads(rng, gb, go, nreads, readlen, ragged=ragged, lower=True, present=np.arange(8))
    if with_n:
        rb[rng.integers(0, rb.size, size=200)] = ord("N")
    return gb, go, rb, ro


def _fastq(rb, ro, crlf=False, final_newline=True):
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i in range(len(ro) - 1):
        seq = bytes(rb[int(ro[i]): int(ro[i + 1])])
        qual = (b"@" if i % 7 == 0 else b"I") * len(seq)  # quality lines that begin with '@': only line COUNTS find records
        out.append(b"@r%d some text" % i + nl + seq + nl + b"+" + nl + qual + nl)
    text = b"".join(out)
    return text if final_newline else text[: -len(nl)]


def _fasta_ml(rb, ro, width=60):
    out = [b"junk in front of the first header\n"]
    for i in range(len(ro) - 1):
        seq = bytes(rb[int(ro[i]): int(ro[i + 1])])
        out.append(b">r%d desc\n" % i)
        for a in range(0, len(seq), width):
            out.append(b"  " + seq[a: a + width] + b" \n")
        if i % 11 == 0:
            out.append(b"\n")
    return b"".join(out)


def _bgzf(data, block=0xff00):
    """BGZF as bgzip writes it: gzip members of <= 64 KB with the BC extra field (BSIZE), then the empty EOF block."""
    out = []
    for a in list(range(0, len(data), block)) + [None]:
        piece = b"" if a is None else data[a: a + block]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = co.compress(piece) + co.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out.append(b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
                   + body + struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece)))
    return b"".join(out)


def _tables(hip, gb, go, ks):
    tabs = [hip.sketch_genomes(gb, go, k, 300)[0] for k in ks]
    return tabs, [int(t.max()) for t in tabs], [hip.filter_build(t) for t in tabs]


def _whole(hip, rb, ro, ks, hmaxs, filts, s=0):
    d_b, d_o = hip.array(rb if rb.size else np.zeros(1, np.uint8)), hip.array(ro)
    sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, len(ro) - 1, ks, hmaxs, s, filts)
    out = []
    for sk in sks:
        sk.resolve()
        out.append(sk.download())
        sk.free()
    d_b.free()
    d_o.free()
    return out


def _finish(stream):
    sks = stream.finish()
    got = []
    try:
        for sk in sks:
            sk.resolve()
            got.append(sk.download())
    finally:
        for sk in sks:
            sk.free()
        stream.free()
    return got


def _streamed(hip, ks, hmaxs, filts, expect, feed):
    """feed(stream) streams the sample; a table sized from the previous sample's distinct-count ratio may prove too small
    (MG_ERR_CAPACITY at resolution, the hint reset by the library): stream again, as select_db.stream_reads_file does."""
    for attempt in range(2):
        stream = hip.sketch_stream(ks, hmaxs, 0, filts, expect_bases=expect)
        feed(stream)
        counts = (stream.nreads, stream.nbases)
        try:
            return _finish(stream), counts
        except _hip.HipError as e:
            if e.code != _hip.ERR_CAPACITY or attempt:
                raise


def _same(got, want):
    assert len(got) == len(want)
    for (gh, gc), (wh, wc) in zip(got, want):
        assert np.array_equal(gh, wh) and np.array_equal(gc, wc)


@pytest.mark.parametrize("ks", [[21, 31, 51], [30, 40, 50, 60], [21], [17, 33]])
def test_pieces_into_one_table_equal_the_whole(hip, oracle_lib, ks):
    gb, go, rb, ro = _sample(1, ragged=True, with_n=True)
    tabs, hmaxs, filts = _tables(hip, gb, go, ks)
    for use_filter in (True, False):
        f = filts if use_filter else None
        want = _whole(hip, rb, ro, ks, hmaxs, f)
        cuts = [0, 1, 64, 65, 9000, 9000, 20000, len(ro) - 1]  # (an empty piece among them)

        def feed(stream):
            for a, b in zip(cuts, cuts[1:]):
                piece = rb[int(ro[a]): int(ro[b])]
                d_b, d_o = hip.array(piece if piece.size else np.zeros(1, np.uint8)), hip.array(ro[a: b + 1] - ro[a])
                stream.add_dev(d_b.ptr, d_o.ptr, b - a, piece.size)
                d_b.free()  # stream-ordered: the hashing kernel queued above still reads it
                d_o.free()
        got, counts = _streamed(hip, ks, hmaxs, f, rb.size, feed)
        assert counts == (len(ro) - 1, rb.size)
        _same(got, want)
    # ... and the oracle, for the first k (unfiltered; saturating counts)
    oh, oc, _, _ = oracle_lib.sketch_reads(rb, ro, ks[0], hmax=hmaxs[0])
    assert np.array_equal(got[0][0], oh) and np.array_equal(got[0][1], oc)


@pytest.mark.parametrize("thin", [False, True])
@pytest.mark.parametrize("chunk", [1 << 16, 3 << 16, 64 << 20])
def test_files_through_the_pipeline(hip, tmp_path, chunk, thin, monkeypatch, knobs):
    """FASTQ (LF, CRLF, no final newline), wrapped FASTA, gzip (one member, several members), BGZF: every form of the same
    reads through mg_sketch_stream_add_file in pieces of `chunk` bytes == the whole reads in one launch.  thin: the plain
    FASTQ files thinned by the reader threads (knob stream_thin: a record goes up as ">" + its sequence line)."""
    if thin:
        knobs("stream_thin", 1)
    ks = [21, 31, 51]
    gb, go, rb, ro = _sample(2, nreads=20000)
    tabs, hmaxs, filts = _tables(hip, gb, go, ks)
    want = _whole(hip, rb, ro, ks, hmaxs, filts)
    fq = _fastq(rb, ro)
    forms = {
        "plain.fq": ("fastq", fq),
        "crlf.fq": ("fastq", _fastq(rb, ro, crlf=True)),
        "nofinal.fq": ("fastq", _fastq(rb, ro, final_newline=False)),
        "trailing_blank.fq": ("fastq", fq + b"\n\n"),
        "wrapped.fa": ("fasta_ml", _fasta_ml(rb, ro)),
        "one.fq.gz": ("fastq", gzip.compress(fq, 4)),
        "members.fq.gz": ("fastq", b"".join(gzip.compress(fq[a: a + 700001], 1) for a in range(0, len(fq), 700001))),
        # trailing padding after the last member (tar / tape blocks): gzip and zcat warn and go on, so does the reader
        "padded.fq.gz": ("fastq", gzip.compress(fq, 4) + b"\0" * 1000),
        "members_padded.fq.gz": ("fastq", b"".join(gzip.compress(fq[a: a + 900001], 1) for a in range(0, len(fq), 900001)) + b"\0\0\0garbage"),
        "bgzf.fq.gz": ("fastq", _bgzf(fq)),
        "bgzf.fa.gz": ("fasta_ml", _bgzf(_fasta_ml(rb, ro))),
    }
    for name, (fmt, data) in forms.items():
        if thin and not name.endswith(".fq"):
            continue  # (only plain FASTQ is thinned)
        p = tmp_path / name
        p.write_bytes(data)
        for threads in (1, 5):
            got, counts = _streamed(hip, ks, hmaxs, filts, rb.size,
                                    lambda st: st.add_file(str(p), fmt, chunk_bytes=chunk, nthreads=threads))
            assert counts == (len(ro) - 1, rb.size), name
            _same(got, want)


def test_byte_range_of_a_plain_file_and_empty_files(hip, tmp_path):
    ks = [21]
    gb, go, rb, ro = _sample(3, nreads=5000)
    tabs, hmaxs, filts = _tables(hip, gb, go, ks)
    fq = _fastq(rb, ro)
    p = tmp_path / "x.fq"
    p.write_bytes(fq)
    # the second half of the records, by byte range (what a rank of a multi-GPU launch is handed)
    first = 2500
    nl = [i for i, ch in enumerate(fq) if ch == 10]
    start = nl[4 * first - 1] + 1
    want = _whole(hip, rb[int(ro[first]):], ro[first:] - ro[first], ks, hmaxs, filts)
    got, counts = _streamed(hip, ks, hmaxs, filts, rb.size, lambda st: st.add_file(str(p), "fastq", offset=start, chunk_bytes=1 << 16))
    assert counts[0] == len(ro) - 1 - first
    _same(got, want)
    for name, data in (("empty.fq", b""), ("empty.fq.gz", gzip.compress(b"")), ("blank.fq", b"\n\n")):
        q = tmp_path / name
        q.write_bytes(data)
        stream = hip.sketch_stream(ks, hmaxs, 0, filts, expect_bases=1)
        stream.add_file(str(q), "fastq")
        assert stream.nreads == 0
        sks = stream.finish()
        sks[0].resolve()
        assert sks[0].size == 0
        sks[0].free()
        stream.free()


@pytest.mark.parametrize("thin", [False, True])
def test_what_the_pipeline_refuses(hip, tmp_path, thin, monkeypatch, knobs):
    if thin:
        knobs("stream_thin", 1)  # (the readers' own checks then: same codes)
    ks = [21]
    gb, go, rb, ro = _sample(4, nreads=2000)
    tabs, hmaxs, filts = _tables(hip, gb, go, ks)
    fq = bytearray(_fastq(rb, ro))
    # a malformed record somewhere in the middle: the device parser's error, as on the whole file
    bad = bytes(fq).replace(b"\n+\n", b"\n-\n", 700).replace(b"\n-\n", b"\n+\n", 699)
    p = tmp_path / "bad.fq"
    p.write_bytes(bad)
    stream = hip.sketch_stream(ks, hmaxs, 0, filts, expect_bases=rb.size)
    with pytest.raises(_hip.HipError) as e:
        stream.add_file(str(p), "fastq", chunk_bytes=1 << 16)
    assert e.value.code == _hip.ERR_ARG
    stream.free()
    # a truncated gzip stream
    g = tmp_path / "cut.fq.gz"
    g.write_bytes(gzip.compress(bytes(_fastq(rb, ro)))[:-40000])
    stream = hip.sketch_stream(ks, hmaxs, 0, filts, expect_bases=rb.size)
    with pytest.raises(_hip.HipError):
        stream.add_file(str(g), "fastq", chunk_bytes=1 << 16)
        stream.finish()
    stream.free()
    # a record longer than a chunk's headroom: MG_ERR_CAPACITY (select_db then takes the piece-wise path)
    fa = tmp_path / "long.fa"
    fa.write_bytes(b">a\n" + b"ACGT" * 40000 + b"\n>b\nACGT\n")
    stream = hip.sketch_stream(ks, hmaxs, 0, filts, expect_bases=200000)
    with pytest.raises(_hip.HipError) as e:
        stream.add_file(str(fa), "fasta_ml", chunk_bytes=1 << 16)
    assert e.value.code == _hip.ERR_CAPACITY
    stream.free()
    # ... and a FASTQ record like that
    lq = tmp_path / "long.fq"
    lq.write_bytes(b"@a\n" + b"ACGT" * 40000 + b"\n+\n" + b"I" * 160000 + b"\n@b\nACGT\n+\nIIII\n")
    stream = hip.sketch_stream(ks, hmaxs, 0, filts, expect_bases=200000)
    with pytest.raises(_hip.HipError) as e:
        stream.add_file(str(lq), "fastq", chunk_bytes=1 << 16)
    assert e.value.code == _hip.ERR_CAPACITY
    stream.free()
    # an undersized table (the estimate was far too small): reported at resolution, the hint reset
    _hip.debug_set("distinct_hint_ppm", 20)
    try:
        stream = hip.sketch_stream(ks, hmaxs, 0, None, expect_bases=rb.size)
    finally:
        _hip.debug_set("distinct_hint_ppm", 0)
    d_b, d_o = hip.array(rb), hip.array(ro)
    stream.add_dev(d_b.ptr, d_o.ptr, len(ro) - 1, rb.size)
    sks = stream.finish()
    with pytest.raises(_hip.HipError) as e:
        sks[0].resolve()
    assert e.value.code == _hip.ERR_CAPACITY
    sks[0].free()
    stream.free()
    d_b.free()
    d_o.free()


@pytest.mark.parametrize("thin", [False, True])
def test_command_lines_streamed_equal_unstreamed(hip, tmp_path, monkeypatch, knobs, thin):
    """select_main and map_main on files: the streamed pipelines (small chunks, so that every file is many pieces with
    carried records / lines) against round 2's whole-file path (MG_NO_STREAM=1) — CSV, subset db_info and CAMI file byte
    for byte; the reads as plain FASTQ, gzip and BGZF.  thin: the plain FASTQ and the SAM file thinned by the reader threads
    (knob stream_thin; the tokeniser takes len(SEQ) from the mark the readers leave)."""
    import argparse

    import test_pipeline_gpu as tp
    from metalign_amd import build_db, map_and_profile, select_db
    rng = np.random.default_rng(77)
    data, gb, go, names, accs = tp._make_data_dir(tmp_path, rng)
    build_db.build([str(data / "organism_files" / nm) for nm in names], str(data / "sketch_table"), [21, 31, 51], 150)
    rb, ro, src = util.sample_reads(rng, gb, go, 6000, 150, err=0.005, present=[3, 8])
    fq = _fastq(rb, ro)
    files = {"s.fq": fq, "s.fq.gz": gzip.compress(fq, 1), "s.bgzf.fq.gz": _bgzf(fq)}
    sam = tmp_path / "a.sam"
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\n")
        for i in range(len(ro) - 1):
            s = bytes(rb[int(ro[i]):int(ro[i + 1])]).decode()
            fh.write("\t".join(["r%d" % i, "0", accs[src[i]], "1", "60", "150M", "*", "0", "0", s, "I" * 150, "NM:i:0"]) + "\n")
            if i % 5 == 0:
                fh.write("\t".join(["r%d" % i, "256", accs[8 if src[i] == 3 else 3], "1", "0", "140M10S", "*", "0", "0", "*", "*", "NM:i:4"]) + "\n")

    def select(reads, tag):
        tmpd = tmp_path / ("tmp_" + tag)
        args = argparse.Namespace(reads=str(reads), data=str(data), cmash_results="NONE", cutoff=0.01, db="AUTO", db_dir="AUTO",
                                  dbinfo_in="AUTO", dbinfo_out="AUTO", input_type="AUTO", keep_temp_files=True, strain_level=False,
                                  temp_dir=str(tmpd), threads=4, sketch_table="AUTO", min_count=2, sketch_size=0)
        select_db.select_main(args)
        return (tmpd / "cmash_query_results.csv").read_text(), (tmpd / "subset_db_info.txt").read_text(), tmpd / "subset_db_info.txt"

    def profile(infile, dbinfo, tag):
        out = tmp_path / ("ab_%s.tsv" % tag)
        a2 = argparse.Namespace(infiles=[str(infile)], data=str(data), db="NONE", dbinfo=str(dbinfo), input_type="AUTO",
                                length_normalize=False, low_mem=False, min_abundance=1e-4, rank_renormalize=False, output=str(out),
                                pct_id=0.5, no_quantify_unmapped=False, read_cutoff=1, sampleID="x", threads=4, verbose=False)
        map_and_profile.map_main(a2)
        return out.read_text()

    monkeypatch.setenv("MG_NO_STREAM", "1")
    p = tmp_path / "s.fq"
    p.write_bytes(fq)
    want_csv, want_sub, sub_path = select(p, "whole")
    want_cami = profile(sam, sub_path, "whole")
    assert want_csv.count("\n") >= 3 and want_cami.count("\n") > 10
    monkeypatch.delenv("MG_NO_STREAM")
    if thin:
        knobs("stream_thin", 1)
    for chunk in ("65536", "0"):
        monkeypatch.setenv("MG_STREAM_CHUNK_BYTES", chunk)
        for name, blob in files.items():
            q = tmp_path / name
            q.write_bytes(blob)
            csv, sub, _ = select(q, "st_%s_%s" % (chunk, name.replace(".", "_")))
            assert (csv, sub) == (want_csv, want_sub), (name, chunk)
        assert profile(sam, sub_path, "st" + chunk) == want_cami


def test_repeated_streams_hold_no_more_memory(hip, tmp_path):
    """Forty streams of the same file (alternating piece sizes, so that the page-locked slots are re-made; a refused file in
    between): the sketches stay the same and the device memory in use does not grow."""
    ks = [21, 31, 51]
    gb, go, rb, ro = _sample(5, nreads=8000)
    tabs, hmaxs, filts = _tables(hip, gb, go, ks)
    p = tmp_path / "x.fq"
    p.write_bytes(_fastq(rb, ro))
    bad = tmp_path / "bad.fq"
    bad.write_bytes(b"@r\nACGT\nnot a plus line\nIIII\n" * 50)
    want = None
    used = []
    for i in range(40):
        got, counts = _streamed(hip, ks, hmaxs, filts, rb.size, lambda st: st.add_file(str(p), "fastq", chunk_bytes=(1 << 16) << (i % 3), nthreads=1 + i % 4))
        assert counts == (len(ro) - 1, rb.size)
        if want is None:
            want = got
        _same(got, want)
        if i % 10 == 5:
            st = hip.sketch_stream(ks, hmaxs, 0, filts, expect_bases=1000)
            with pytest.raises(_hip.HipError):
                st.add_file(str(bad), "fastq", chunk_bytes=1 << 16)
            st.free()
        hip.sync()
        free, total, pooled = hip.mem_info()
        used.append(total - free - pooled)
    assert max(used[20:]) <= max(used[5:20]) + (8 << 20), used
