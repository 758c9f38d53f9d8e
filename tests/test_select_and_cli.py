"""CPU: the pre-filter's host logic and the three command lines against the reference's golden vectors."""
import argparse
import json
import os

import pytest

from metalign_amd import map_and_profile, metalign, select_db

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEL = os.path.join(GOLDEN, "select")


def _capture_parser(fn):
    got = []

    class Stop(Exception):
        pass

    def fake(self, *a, **k):
        got.append(self)
        raise Stop()

    real = argparse.ArgumentParser.parse_args
    argparse.ArgumentParser.parse_args = fake
    try:
        try:
            fn()
        except Stop:
            pass
    finally:
        argparse.ArgumentParser.parse_args = real
    return got[0]


@pytest.mark.parametrize("name,fn", [("metalign", metalign.metalign_parseargs),
                                     ("select_db", select_db.select_parseargs),
                                     ("map_and_profile", map_and_profile.profile_parseargs)])
def test_cli_flags_match_reference(name, fn):
    """Every flag of the reference exists with the same default / type / choices / arity; the build's own
    additions are optional flags only."""
    with open(os.path.join(GOLDEN, "cli_flags.json")) as fh:
        want = json.load(fh)[name]
    parser = _capture_parser(fn)
    mine = {a.dest: a for a in parser._actions if not isinstance(a, argparse._HelpAction)}
    for f in want:
        a = mine.pop(f["dest"])
        assert list(a.option_strings) == f["options"]
        assert a.default == f["default"], f["dest"]
        assert getattr(a.type, "__name__", None) == f["type"]
        assert (list(a.choices) if a.choices else None) == f["choices"]
        assert a.nargs == f["nargs"]
        assert type(a).__name__ == f["action"]
    for extra in mine.values():
        assert extra.option_strings, "build-only additions must be optional flags: %s" % extra.dest


@pytest.mark.parametrize("case", ["basic", "all_below", "single_k"])
def test_select_from_cmash_csv_matches_reference(case, tmp_path):
    with open(os.path.join(SEL, case + ".json")) as fh:
        runs = json.load(fh)
    for run in runs:
        args = argparse.Namespace(reads="reads.fq", data=SEL + "/", cmash_results=os.path.join(SEL, case + ".csv"),
                                  cutoff=0.01, db=str(tmp_path / "db.fna"), db_dir="AUTO", dbinfo_in="AUTO",
                                  dbinfo_out=str(tmp_path / "sub.txt"), input_type="AUTO", keep_temp_files=False,
                                  strain_level=False, temp_dir=str(tmp_path) + "/", threads=1)
        for k, v in run["args"].items():
            setattr(args, k, v)
        select_db.select_main(args)
        t2i = select_db.read_dbinfo(args)
        assert select_db.run_cmash_and_cutoff(args, t2i) == run["organisms"]
        assert open(args.dbinfo_out).read() == run["subset_dbinfo"]
        assert open(args.db).read() == run["fasta"]


def test_cutoff_out_of_range_exits(capsys):
    args = argparse.Namespace(cutoff=1.5)
    with pytest.raises(SystemExit):
        select_db.select_main(args)
    assert "between 0 and 1" in capsys.readouterr().out


def test_containment_csv_layout(tmp_path):
    import numpy as np
    names = ["taxid_1_1_genomic.fna.gz", "taxid_2_1_genomic.fna.gz", "taxid_3_1_genomic.fna.gz", "taxid_4_1_genomic.fna.gz"]
    per_k = [np.array([0.5, 0.0, 0.25, 0.1]), np.array([0.2, 0.0, 0.9, 0.2])]
    rows = select_db.containment_rows(names, per_k)
    assert [r[0] for r in rows] == [names[2], names[0], names[3]]  # zero-at-smallest-k dropped; ties keep table order
    p = tmp_path / "q.csv"
    select_db.write_containment_csv(str(p), [21, 31], rows)
    lines = p.read_text().splitlines()
    assert lines[0] == ",k=21,k=31"
    assert lines[1] == names[2] + ",0.25,0.9"
    assert float(lines[2].split(",")[-1]) == 0.2


def test_fasta_fastq_readers(tmp_path):
    import gzip

    from metalign_amd import formats
    fa = tmp_path / "x.fa"
    fa.write_text(">a desc\nACGT\nNNac\n>b\n\n>c\nTT\n")
    b, o, n = formats.read_sequences(str(fa), "fasta")
    assert n == ["a", "b", "c"] and list(o) == [0, 8, 8, 10] and bytes(b) == b"ACGTNNacTT"
    fq = tmp_path / "x.fq.gz"
    with gzip.open(str(fq), "wt") as fh:
        fh.write("@r1 x\nACG\n+\nIII\n@r2\nTTTT\n+\nIIII\n")
    b, o, n = formats.read_sequences(str(fq), "fastq")
    assert n == ["r1", "r2"] and list(o) == [0, 3, 7] and bytes(b) == b"ACGTTTT"


def test_paf_adaptor_maps_onto_sam_records():
    """A SAM stream and its PAF rendering (with cg:Z:) give the same records wherever PAF can express them."""
    import numpy as np

    from metalign_amd import map_and_profile as mp
    accs = ["NZ_A.1", "NZ_B.1"]
    idx = {"Unmapped": 0, "NZ_A.1": 1, "NZ_B.1": 2}
    sam = ("r1\t0\tNZ_A.1\t10\t60\t5S35M\t*\t0\t0\t" + "A" * 40 + "\t" + "I" * 40 + "\tNM:i:1\n"
           "r1\t272\tNZ_B.1\t99\t0\t30M2D8M2S\t*\t0\t0\t*\t*\tNM:i:4\n"
           "r2\t16\tNZ_B.1\t5\t60\t40M\t*\t0\t0\t" + "C" * 40 + "\t" + "I" * 40 + "\tNM:i:0\n")
    paf = ("r1\t40\t5\t40\t+\tNZ_A.1\t5000\t9\t44\t34\t35\t60\ttp:A:P\tcg:Z:35M\n"
           "r1\t40\t0\t38\t-\tNZ_B.1\t5000\t98\t138\t36\t40\t0\ttp:A:S\tcg:Z:30M2D8M\n"
           "r2\t40\t0\t40\t-\tNZ_B.1\t5000\t4\t44\t40\t40\t60\ttp:A:P\tcg:Z:40M\n")
    a = mp.tokenise_sam(sam.splitlines(True), idx)
    b = mp.tokenise_paf(paf.splitlines(True), idx)
    assert np.array_equal(a, b)
    # without cg:Z: the identity test is approximated by matching bases / query length
    c = mp.tokenise_paf(["r3\t100\t0\t100\t+\tNZ_A.1\t5000\t0\t100\t90\t100\t60\ttp:A:P\n"], idx)
    assert (int(c["matched"][0]), int(c["total"][0])) == (90, 100)


def test_sketch_table_v2_hash_major_round_trip(tmp_path):
    """formats: the hash-major table (version 2) — what the builder writes is what stage B streams; slices by hash range
    touch only their part and add up; the genome-major view and version-1 directories still read the same table."""
    import numpy as np
    from metalign_amd import formats
    rng = np.random.default_rng(4)
    G, n = 7, 50
    per_k = {}
    for k in (21, 31):
        sk = [np.sort(rng.choice(1 << 40, size=int(rng.integers(0, n + 1)), replace=False).astype(np.uint64)) for _ in range(G)]
        sk[3] = sk[2][: len(sk[2]) // 2].copy()  # shared hashes between genomes
        o = np.zeros(G + 1, dtype=np.uint64)
        o[1:] = np.cumsum([len(x) for x in sk])
        per_k[k] = (np.concatenate(sk), o)
    names = ["taxid_%d_1_genomic.fna.gz" % g for g in range(G)]
    filters = {21: np.arange(2048, dtype=np.uint32)}
    formats.write_sketch_table(str(tmp_path / "v2"), names, [21, 31], n, per_k, filters)
    formats.write_sketch_table_v1(str(tmp_path / "v1"), names, [21, 31], n, per_k)
    for d in ("v2", "v1"):
        t = formats.SketchTable(str(tmp_path / d))
        assert t.ks == [21, 31] and t.ngenomes == G and t.names == names
        for k in (21, 31):
            h, o = t.arrays(k)
            assert np.array_equal(h, per_k[k][0]) and np.array_equal(o, per_k[k][1])
            full = t.pairs(k)
            ph, pg = np.asarray(full["pair_hash"]), np.asarray(full["pair_gen"])
            assert np.all(ph[1:] >= ph[:-1]) and len(ph) == len(per_k[k][0]) and full["max_hash"] == int(per_k[k][0].max())
            assert np.array_equal(full["gsize"], np.diff(per_k[k][1]).astype(np.uint32))
            same = ph[1:] == ph[:-1]
            assert np.all(pg[1:][same] > pg[:-1][same])  # equal hashes: genome order
            for g in range(G):  # every genome's pairs are its sketch
                assert np.array_equal(ph[pg == g], per_k[k][0][int(per_k[k][1][g]):int(per_k[k][1][g + 1])])
            bounds = [0, int(ph[len(ph) // 3]), int(ph[2 * len(ph) // 3]), full["max_hash"] + 1]
            parts = [t.pairs(k, bounds[i], bounds[i + 1]) for i in range(3)]
            assert np.array_equal(np.concatenate([np.asarray(q["pair_hash"]) for q in parts]), ph)
            assert np.array_equal(sum(q["gsize"].astype(np.int64) for q in parts), full["gsize"])
    t2 = formats.SketchTable(str(tmp_path / "v2"))
    assert np.array_equal(t2.filter_bits(21), filters[21]) and t2.filter_bits(31) is None


def test_tokenise_paf_takes_bytes_lines_like_the_stream_code_hands_them(tmp_path):
    """compute_abundances opens replay files in binary mode: PAF lines arrive as bytes (this crashed with TypeError)."""
    import numpy as np
    from metalign_amd import map_and_profile as mp
    idx = {"Unmapped": 0, "NZ_A.1": 1}
    line = "r1\t40\t5\t40\t+\tNZ_A.1\t5000\t9\t44\t34\t35\t60\ttp:A:P\tcg:Z:35M\n"
    a = mp.tokenise_paf([line], idx)
    b = mp.tokenise_paf([line.encode()], idx)
    p = tmp_path / "x.paf"
    p.write_text(line * 3)
    with open(str(p), "rb") as fh:
        c = mp.tokenise_paf(fh, idx, decode=False)
    assert np.array_equal(a, b) and len(c) == 3 and np.array_equal(c[:1], a)


def test_record_aligned_byte_ranges_of_a_multi_gpu_launch(tmp_path):
    """select_db.read_range_of_rank: the byte ranges the ranks of a torch.distributed.run launch take of the reads file
    tile it at RECORD boundaries — FASTQ with quality lines that begin with '@' and '+' (a byte-pattern guess would be
    fooled; the ranges come from newline counts), reads of very different lengths, no final newline; FASTA with wrapped
    sequences and junk in front of the first header — for every world size, every record exactly once."""
    import numpy as np
    from metalign_amd import select_db
    rng = np.random.default_rng(12)
    recs = []
    for i in range(257):
        n = int(rng.integers(1, 400))
        seq = "".join(rng.choice(list("ACGTN"), size=n))
        qual = "".join(rng.choice(list("@+I#>"), size=n))
        if i % 3 == 0:
            qual = "@" + qual[1:]
        if i % 5 == 0:
            qual = "+" + qual[1:]
        recs.append("@r%d +x\n%s\n+\n%s\n" % (i, seq, qual))
    for tail in ("", "strip"):
        text = "".join(recs)
        if tail:
            text = text[:-1]  # no newline at the end of the file
        fq = tmp_path / ("r%s.fq" % tail)
        fq.write_text(text)
        size = len(text)
        for world in range(1, 9):
            raw = [size * r // world for r in range(world + 1)]
            counts = [select_db.count_newlines(str(fq), raw[r], raw[r + 1]) for r in range(world)]
            ranges = [select_db.read_range_of_rank(str(fq), "fastq", r, world, lambda x: counts) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == size
            assert all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
            starts = set(np.cumsum([0] + [len(x) for x in recs[:-1]]).tolist()) | {size}
            assert all(a in starts and b in starts and a <= b for a, b in ranges), (world, ranges)
    fa_recs = [">s%d desc > not a header\n%s\n" % (i, "\n".join("".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 61))))
                                                              for _ in range(int(rng.integers(0, 6))))) for i in range(120)]
    text = "junk in front\n\n" + "".join(fa_recs)
    fa = tmp_path / "r.fa"
    fa.write_text(text)
    size = len(text)
    hdr = set((len("junk in front\n\n") + np.cumsum([0] + [len(x) for x in fa_recs[:-1]])).tolist()) | {0, size}
    for world in range(1, 9):
        ranges = [select_db.read_range_of_rank(str(fa), "fasta", r, world, None) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == size
        assert all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
        assert all(a in hdr and b in hdr and a <= b for a, b in ranges), (world, ranges)


def test_sam_pieces_of_a_multi_gpu_launch_give_the_same_record_stream(tmp_path):
    """map_and_profile under torch.distributed.run: the ranks tokenise line-aligned byte ranges of the SAM file
    independently (every piece's first record gets its new-read bit) and rank 0 clears that bit wherever a piece's
    first retained QNAME equals the last retained QNAME in front of it (clear_continued_heads).  Emulated here with the
    host tokeniser on the pieces: for every world size the concatenation equals the tokenisation of the whole file —
    with reads of several lines straddling the cuts, header and unmapped lines at the cuts, empty pieces."""
    import numpy as np
    from metalign_amd import _hip
    from metalign_amd import map_and_profile as mp
    rng = np.random.default_rng(5)
    accs = ["ACC%03d.1" % i for i in range(20)]
    idx = {a: i for i, a in enumerate(accs)}
    lines = ["@HD\tVN:1.6", "@SQ\tSN:ACC000.1\tLN:1000"]
    for r in range(400):
        nl = int(rng.integers(1, 6))
        for j in range(nl):
            flag = 0 if j == 0 else 256
            if rng.random() < 0.1:
                lines.append("q%d\t4\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII" % r)  # unmapped: filtered, same QNAME
            seq = "ACGT" * int(rng.integers(5, 40)) if j == 0 else "*"
            lines.append("\t".join(["q%d" % r, str(flag), accs[int(rng.integers(0, 20))], "1", "60", "%dM" % (len(seq) if j == 0 else 50),
                                     "*", "0", "0", seq, "I" * len(seq) if j == 0 else "*", "NM:i:0"]))
        if r % 50 == 7:
            lines.append("@CO\ta comment in the middle")
    text = "\n".join(lines) + "\n"
    path = tmp_path / "a.sam"
    path.write_text(text)
    whole = mp.tokenise_sam(text.splitlines(True), idx)
    size = len(text)
    for world in (1, 2, 3, 5, 8, 64, 900):
        ranges = [mp.sam_range_of_rank(str(path), r, world) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == size and all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
        assert all(a == 0 or text[a - 1] == "\n" for a, _ in ranges)
        pieces, firsts, lasts = [], [], []
        for a, b in ranges:
            tk = mp._Tokeniser(idx)
            for ln in text[a:b].splitlines(True):
                tk.feed(ln)
            pieces.append(tk.records())
            firsts.append(mp.first_retained_qname(str(path), a, b) or "")
            lasts.append(tk.prev)
        got = np.concatenate(pieces) if pieces else np.zeros(0, dtype=_hip.REC_DTYPE)

        def clear(i):
            got["ref_new"][i] &= 0x7FFFFFFF
        offs = mp.clear_continued_heads(clear, [len(p) for p in pieces], firsts, lasts)
        assert offs[-1] == len(whole) and np.array_equal(got, whole), world


def test_only_rank_0_writes_the_subset_database_when_cmash_results_is_given(tmp_path, monkeypatch):
    """A torch.distributed.run launch with --cmash_results: nothing to sketch, and the host-only tail
    (run_cmash_and_cutoff + make_db_and_dbinfo, reference scripts/select_db.py:80-117) has ONE writer — W ranks
    appending zcat output to the same args.db would interleave."""
    def args_for(sub):
        d = tmp_path / sub
        d.mkdir()
        return argparse.Namespace(reads="reads.fq", data=SEL + "/", cmash_results=os.path.join(SEL, "basic.csv"), cutoff=0.01,
                                  db=str(d / "db.fna"), db_dir="AUTO", dbinfo_in="AUTO", dbinfo_out=str(d / "sub.txt"),
                                  input_type="fastq", keep_temp_files=False, strain_level=False, temp_dir=str(d) + "/", threads=1)
    monkeypatch.setenv("WORLD_SIZE", "4")
    monkeypatch.setenv("RANK", "2")
    a = args_for("rank2")
    select_db.select_main(a)
    assert not os.path.exists(a.db) and not os.path.exists(a.dbinfo_out)
    monkeypatch.setenv("RANK", "0")
    a = args_for("rank0")
    select_db.select_main(a)
    assert os.path.getsize(a.db) > 0 and os.path.getsize(a.dbinfo_out) > 0


def test_other_ranks_leave_a_shared_temp_dir_alone(tmp_path, monkeypatch):
    """metalign.py under torch.distributed.run with an explicit --temp_dir: the directory is shared by all ranks and
    rank 0 is still working in it (CSV, subset database, the aligner) when the others are done — only rank 0 removes
    it, after map_main; with the AUTO default every rank removes the directory it made itself."""
    calls = []
    monkeypatch.setattr(select_db, "select_main", lambda args: calls.append(("select", args.temp_dir)))
    monkeypatch.setattr(map_and_profile, "map_main", lambda args: calls.append(("map", args.temp_dir)))
    monkeypatch.setenv("WORLD_SIZE", "2")
    shared = tmp_path / "shared"
    shared.mkdir()
    (shared / "cmash_query_results.csv").write_text("x")
    data = tmp_path / "data"
    data.mkdir()
    argv = ["reads.fq", str(data), "--temp_dir", str(shared), "--input_type", "fastq"]
    monkeypatch.setenv("RANK", "1")
    metalign.main(argv)
    assert shared.exists() and (shared / "cmash_query_results.csv").exists() and [c[0] for c in calls] == ["select"]
    monkeypatch.setenv("RANK", "0")
    metalign.main(argv)
    assert not shared.exists() and [c[0] for c in calls] == ["select", "select", "map"]
    # AUTO: a rank's own mkdtemp goes away with the rank
    calls.clear()
    monkeypatch.setenv("RANK", "1")
    metalign.main(["reads.fq", str(data), "--input_type", "fastq"])
    assert calls and not os.path.exists(calls[0][1])


def test_inflate_file_takes_every_member_and_refuses_a_truncated_stream(tmp_path):
    """`.gz` reads are expected input (reference scripts/select_db.py:146-148): formats.inflate_file is what rank 0 of a
    multi-GPU launch inflates them with."""
    import gzip

    from metalign_amd import formats
    data = bytes(range(256)) * 4000 + b"ACGT" * 50000
    one, two, cut, empty = (tmp_path / n for n in ("one.gz", "two.gz", "cut.gz", "empty.gz"))
    one.write_bytes(gzip.compress(data, 1))
    two.write_bytes(gzip.compress(data[:300000], 1) + gzip.compress(data[300000:], 9))
    cut.write_bytes(gzip.compress(data, 1)[:-3000])
    empty.write_bytes(gzip.compress(b""))
    assert formats.is_gzip(str(one)) and not formats.is_gzip(__file__)
    assert formats.inflate_file(str(one)) == data
    assert formats.inflate_file(str(two), block=70001) == data
    assert formats.inflate_file(str(empty)) == b""
    with pytest.raises(OSError):
        formats.inflate_file(str(cut))


def test_subset_database_is_what_zcat_would_write(tmp_path, capfd):
    """make_db_and_dbinfo writes `zcat <organism file>` per selected genome (reference scripts/select_db.py:103-105, exit
    codes ignored): gzip files — every member, padding after the last ignored — inflated in order by the library's host
    threads (mg_zcat_files); a file zcat refuses (not gzip, cut short, missing) contributes nothing but a message."""
    import gzip

    import numpy as np
    from metalign_amd import formats

    org = tmp_path / "org"
    org.mkdir()
    (org / "a.fna.gz").write_bytes(gzip.compress(b">a\nACGT\n"))
    (org / "b.fna.gz").write_bytes(gzip.compress(b">b1\nAA\n") + gzip.compress(b">b2\nCC\n") + b"\0" * 512)
    (org / "c.fna.gz").write_bytes(b">not gzip\nGG\n")
    (org / "d.fna.gz").write_bytes(gzip.compress(b">d\nTT\n"))
    (org / "e.fna.gz").write_bytes(gzip.compress(b">e\n" + b"ACGT" * 5000 + b"\n")[:-9])
    (org / "f.fna.gz").write_bytes(b"")
    big = b">g\n" + bytes(np.random.default_rng(3).choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=300000)) + b"\n"
    (org / "g.fna.gz").write_bytes(gzip.compress(big))
    out = tmp_path / "db.fna"
    out.write_bytes(b"left over from an earlier run")
    names = ("a.fna.gz", "b.fna.gz", "c.fna.gz", "d.fna.gz", "e.fna.gz", "f.fna.gz", "missing.fna.gz", "g.fna.gz")
    select_db._zcat_into(str(out), [str(org / n) for n in names], threads=3)
    assert out.read_bytes() == b">a\nACGT\n>b1\nAA\n>b2\nCC\n>d\nTT\n" + big
    err = capfd.readouterr().err
    for bad in ("c.fna.gz", "e.fna.gz", "missing.fna.gz"):
        assert "zcat: " in err and "/" + bad in err
    for good in ("a.fna.gz", "b.fna.gz", "f.fna.gz", "g.fna.gz"):
        assert "/" + good not in err
    # the same bytes as the Python inflater gives (formats.inflate_file: what the multi-GPU launch reads `.gz` input with)
    want = b""
    for n in names:
        try:
            want += formats.inflate_file(str(org / n))
        except Exception:
            pass
    assert out.read_bytes() == want
    select_db._zcat_into(str(out), [], threads=2)
    assert out.read_bytes() == b""


class _FakeSketch:
    def __init__(self, fail_with=None):
        self.fail_with, self.freed = fail_with, False

    def resolve(self):
        if self.fail_with is not None:
            raise self.fail_with

    def free(self):
        self.freed = True


class _FakeStreamHip:
    """Stands in for _hip.Hip in select_db.stream_reads_file: records what it is asked and fails as scripted."""

    def __init__(self, script):
        self.script, self.calls = list(script), []

    def sketch_stream(self, ks, hmaxs, s, filts, expect):
        hip = self
        step = self.script.pop(0)

        class S:
            nbases = 123_456_789

            def add_file(self, path, fmt, offset=0, length=0, chunk_bytes=0):
                hip.calls.append(("add_file", fmt, expect))
                if step.get("add"):
                    raise step["add"]

            def finish(self):
                return [_FakeSketch(step.get("resolve")) for _ in ks]

            def free(self):
                hip.calls.append(("free",))
        return S()


def test_stream_reads_file_retries_and_falls_back(tmp_path):
    """The host logic around mg_sketch_stream_*: a counting table that proved too small (MG_ERR_CAPACITY at resolution) ->
    the file is streamed ONCE more, sized for what was seen; a record too long for the pieces, or FASTA text the device
    parser refuses -> None (the caller's piece-wise path); a malformed FASTQ record -> the error surfaces."""
    from metalign_amd import _hip
    fq = tmp_path / "r.fq"
    fq.write_bytes(b"@r\nACGT\n+\nIIII\n" * 10)
    cap = _hip.HipError("overflow", _hip.ERR_CAPACITY)
    arg = _hip.HipError("malformed", _hip.ERR_ARG)
    hip = _FakeStreamHip([{"resolve": cap}, {}])
    sks = select_db.stream_reads_file(hip, str(fq), "fastq", [21], [1], 0, [None])
    assert len(sks) == 1 and [c[0] for c in hip.calls] == ["add_file", "free", "add_file", "free"]
    assert hip.calls[2][2] >= int(123_456_789 * 1.25)  # the second pass is sized for the bases the first one counted
    hip = _FakeStreamHip([{"resolve": cap}, {"resolve": cap}])
    assert select_db.stream_reads_file(hip, str(fq), "fastq", [21], [1], 0, [None]) is None
    hip = _FakeStreamHip([{"add": cap}])
    assert select_db.stream_reads_file(hip, str(fq), "fastq", [21], [1], 0, [None]) is None
    hip = _FakeStreamHip([{"add": arg}])
    assert select_db.stream_reads_file(hip, str(fq), "fasta", [21], [1], 0, [None]) is None and hip.calls[0][1] == "fasta_ml"
    hip = _FakeStreamHip([{"add": arg}])
    with pytest.raises(_hip.HipError):
        select_db.stream_reads_file(hip, str(fq), "fastq", [21], [1], 0, [None])
    assert select_db.expected_bases(str(fq), "fastq") == os.path.getsize(fq) // 2 + 1


def test_when_a_job_builds_the_resident_index(monkeypatch):
    """HipEngine.wants_resident_index: forced by MG_RESIDENT_INDEX, otherwise for tables whose largest hash lets at least
    5 % of all k-mers through (under either hash definition) and whose indexes fit half of the free HBM."""
    from metalign_amd import distributed

    class FakeHip:
        hash_mode = 0

        def __init__(self, free):
            self.free = free

        def mem_info(self):
            return self.free, 288 << 30, 0

    eng = distributed.HipEngine(FakeHip(200 << 30))
    eng.nk = 3
    monkeypatch.delenv("MG_RESIDENT_INDEX", raising=False)
    dense, sparse = int(0.22 * 2 ** 64), int(0.02 * 2 ** 64)
    assert eng.wants_resident_index(dense, 200_000_000)
    assert not eng.wants_resident_index(sparse, 200_000_000)
    assert not eng.wants_resident_index(dense, 2_000_000_000)      # 3 k x 2 copies x 48 B x 2e9 hashes do not fit
    eng.hip.hash_mode = 1                                          # hashes below 9999999999971: the same share of them
    assert eng.wants_resident_index(int(0.22 * 9999999999971), 1000)
    assert not eng.wants_resident_index(int(0.01 * 9999999999971), 1000)
    # ... at half the load (twice the memory) when that fits too
    monkeypatch.delenv("MG_RESIDENT_SPREAD", raising=False)
    assert eng.resident_spread(1000) == 1 and eng.resident_spread(200_000_000) == 0
    eng.hip.free = 280 << 30
    assert eng.resident_spread(200_000_000) == 1
    monkeypatch.setenv("MG_RESIDENT_SPREAD", "0")
    assert eng.resident_spread(1000) == 0
    monkeypatch.setenv("MG_RESIDENT_INDEX", "0")
    assert not eng.wants_resident_index(dense, 1000)
    monkeypatch.setenv("MG_RESIDENT_INDEX", "1")
    assert eng.wants_resident_index(sparse, 1000)
    eng.bottom_s = 500                                             # a bottom-s job: the library would not use the index
    assert not eng.wants_resident_index(dense, 1000)



def test_inflate_file_ignores_trailing_padding_and_refuses_truncation(tmp_path):
    """formats.inflate_file (the subset database's zcat, the multi-GPU launch's reads): every member of a file, trailing
    garbage after the last member ignored as gzip / zcat do, a stream that ends inside a member refused."""
    import gzip
    from metalign_amd import formats
    text = b"@r0\nACGT\n+\nIIII\n" * 5000
    a, b = gzip.compress(text[:30000], 1), gzip.compress(text[30000:], 6)
    for name, blob in (("one.gz", gzip.compress(text)), ("two.gz", a + b), ("padded.gz", a + b + b"\0" * 777), ("junk.gz", a + b + b"garbage")):
        p = tmp_path / name
        p.write_bytes(blob)
        assert formats.inflate_file(str(p), block=4096) == text, name
    p = tmp_path / "cut.gz"
    p.write_bytes((a + b)[:-9])
    with pytest.raises(OSError):
        formats.inflate_file(str(p))



def test_text_record_cuts_for_the_scattered_gz_reads():
    """select_db.text_record_cuts: the record-aligned shares rank 0 scatters after inflating a `.gz` reads file — FASTQ by line
    number (quality lines that begin with '@' or '+'), FASTA by '>' after a newline (a '>' inside a sequence line is not one)."""
    import numpy as np
    from metalign_amd import select_db
    rng = np.random.default_rng(0)
    recs = []
    for i in range(1000):
        ln = int(rng.integers(1, 200))
        seq = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=ln))
        qual = bytes(rng.choice(np.frombuffer(b"@IJ+>", dtype=np.uint8), size=ln))
        recs.append(b"@r%d\n" % i + seq + b"\n+\n" + qual + b"\n")
    text = b"".join(recs)
    starts = set(np.cumsum([0] + [len(r) for r in recs]).tolist())
    for world in (1, 2, 3, 7, 8, 64, 2000):
        cuts = select_db.text_record_cuts(text, "fastq", world)
        assert cuts[0] == 0 and cuts[-1] == len(text) and len(cuts) == world + 1 and cuts == sorted(cuts)
        assert all(c in starts for c in cuts), world
    fa = b"junk\n" + b"".join(b">s%d desc\nACGT\nAC>GT\n" % i for i in range(500))
    for world in (2, 5, 9):
        cuts = select_db.text_record_cuts(fa, "fasta", world)
        assert all(c == 0 or c == len(fa) or (fa[c:c + 1] == b">" and fa[c - 1:c] == b"\n") for c in cuts), (world, cuts)
    assert select_db.text_record_cuts(b"", "fastq", 4) == [0, 0, 0, 0, 0]


def test_build_db_refuses_an_unsupported_k_of_the_cmash_mode_before_reading_a_genome(tmp_path, capsys):
    """Hash mode 1 and the forward sketch hash are built for a list of k (mg_hash_mode1_ks): a k outside it is a command-line
    error naming the list — not an MG_ERR_ARG at the first kernel launch, after the genomes have been read."""
    from metalign_amd import _hip, build_db
    ks = _hip.hash_mode1_ks()
    assert {30, 40, 50, 60, 21, 31, 51} <= set(ks) and ks == sorted(ks)
    listing = tmp_path / "genomes.txt"
    listing.write_text("")
    for argv in (["--hash_mode", "cmash", "-k", "20,28,36"], ["--reference_pipeline", "--sketch_hash", "forward", "-k", "21,31,52"]):
        with pytest.raises(SystemExit):
            build_db.main([str(listing), str(tmp_path / "out")] + argv)
        err = capsys.readouterr().err
        assert "built for k in {1, 5, 10" in err and ("28, 36" in err or "52" in err)


def test_the_verification_kit_runs_dry_and_its_comparison_tells_differences_apart(tmp_path, capsys):
    """tools/verify_against_cmash.sh --dry_run (no KMC, no CMash, no GPU): the fixture is written and checked, every command of
    both sides is printed; tools/verify_kit/compare_csv.py: identical tables (row order aside) pass, a differing last column and a
    row present on one side only are reported with what they would mean."""
    import importlib.util
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "tools", "verify_against_cmash.sh"), "--dry_run", str(tmp_path / "kit")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    for needle in ("kmc -v -k60 -fq -ci2 -cs3", "kmc_tools simple", "intersect", "StreamingQueryDNADatabase.py", "30-60-10 -c 0 -r 1000000 -v -f",
                   "--sensitive", "MakeStreamingDNADatabase.py", "-n 1000 -k 60", "metalign_amd.build_db", "--reference_pipeline --hash_mode cmash",
                   "metalign_amd.select_db", "dry run fine: fixture of 20 genomes"):
        assert needle in r.stdout, needle
    expect = json.load(open(tmp_path / "kit" / "expected.json"))
    assert expect["present"] == [0, 2, 5, 6, 7] and expect["expected_k60"]["3"] == "high" and expect["expected_k60"]["9"] == "low"
    spec = importlib.util.spec_from_file_location("compare_csv", os.path.join(root, "tools", "verify_kit", "compare_csv.py"))
    cmp_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cmp_mod)
    a, b, c = tmp_path / "a.csv", tmp_path / "b.csv", tmp_path / "c.csv"
    a.write_text(",k=30,k=60\n/x/g1.fna.gz,0.9,0.8\n/x/g2.fna.gz,0.5,0.1\n")
    b.write_text(",k=30,k=60\ng2.fna.gz,0.5,0.1\ng1.fna.gz,0.9,0.8\n")
    c.write_text(",k=30,k=60\ng1.fna.gz,0.9,0.7\ng3.fna.gz,0.2,0.0\n")
    assert cmp_mod.main([str(a), str(b)]) == 0
    assert "identical within" in capsys.readouterr().out
    assert cmp_mod.main([str(a), str(c)]) == 1
    out = capsys.readouterr().out
    assert "only in REFERENCE: g2.fna.gz" in out and "only in OURS: g3.fna.gz" in out and "column k=60: 1 organisms differ" in out
    assert "the LAST column (k_max) differs" in out
