"""-m gpu: the PAF replay adaptor on the device.

The reference reads SAM (/root/reference/scripts/map_and_profile.py:87,97,142-144,211,217; minimap2 is run with -a,
:413-415), so PAF has no reference counterpart: parity is defined THROUGH the SAM path — a SAM stream and its PAF
rendering (with cg:Z:) must give the same records, and therefore the same CAMI profile, wherever PAF can express the
SAM (single-end, flags 0/16/256/272)."""
import random
import re

import numpy as np
import pytest

import samgen
import stage_c_checks as sc
from metalign_amd import _hip
from metalign_amd import map_and_profile as mp

pytestmark = pytest.mark.gpu


def _sam_stream(seed, nreads, accs, taxids, readlen=60):
    rng = random.Random(seed)
    by_tax = {}
    for a, t in zip(accs, taxids):
        by_tax.setdefault(t, []).append(a)
    taxa = sorted(by_tax)
    weights = [1.0 / (1 + i) ** 1.3 for i in range(len(taxa))]
    out = []
    for r in range(nreads):
        q = "read%d" % r
        t = rng.choices(taxa, weights)[0]
        strand = 16 if rng.random() < 0.5 else 0
        u = rng.random()
        seq = samgen._seq(rng, readlen)
        cig = samgen._cigar_bad(rng, readlen) if u < 0.06 else samgen._cigar_ok(rng, readlen)
        out.append(samgen._line(q, strand, rng.choice(by_tax[t]), cig, seq, rng.randrange(4)))
        if u > 0.6:  # secondaries (SEQ '*')
            for _ in range(rng.randrange(1, 4)):
                t2 = t if rng.random() < 0.35 else rng.choices(taxa, weights)[0]
                c2 = samgen._cigar_bad(rng, readlen) if rng.random() < 0.15 else samgen._cigar_ok(rng, readlen)
                out.append(samgen._line(q, 256 | strand, rng.choice(by_tax[t2]), c2, "*", rng.randrange(6)))
    return out


def _to_paf(sam_line):
    f = sam_line.rstrip("\n").split("\t")
    flag, cigar = int(f[1]), f[5]
    ops = re.findall(r"(\d+)([A-Z])", cigar)
    qlen = sum(int(n) for n, o in ops if o in "MIS")
    lead = int(ops[0][0]) if ops[0][1] == "S" else 0
    trail = int(ops[-1][0]) if ops[-1][1] == "S" and len(ops) > 1 else 0
    cg = "".join(n + o for n, o in ops if o != "S")
    nmatch = sum(int(n) for n, o in ops if o == "M")
    blen = sum(int(n) for n, o in ops if o in "MID")
    return "\t".join([f[0], str(qlen), str(lead), str(qlen - trail), "-" if flag & 16 else "+", f[2], "50000", "999",
                      str(999 + blen), str(nmatch), str(blen), "0" if flag & 256 else "60",
                      "NM:i:1", "tp:A:" + ("S" if flag & 256 else "P"), "cg:Z:" + cg]) + "\n"


def test_paf_replay_equals_sam_replay(hip, tmp_path):
    dbinfo_text, accs, taxids = samgen.make_dbinfo()
    dbinfo = tmp_path / "db_info.txt"
    dbinfo.write_text(dbinfo_text)
    sam_lines = _sam_stream(3, 20000, accs, taxids)
    paf_lines = [_to_paf(ln) for ln in sam_lines]
    sam, paf = tmp_path / "x.sam", tmp_path / "x.paf"
    sam.write_text("".join(sam_lines))
    paf.write_text("".join(paf_lines))
    acc_index = {"Unmapped": 0}
    acc_index.update({a: i + 1 for i, a in enumerate(accs)})
    # records: device PAF tokeniser == device SAM tokeniser == host PAF tokeniser (chunked through the same stream code)
    from_sam = mp.tokenise_sam_device(open(str(sam), "rb"), acc_index)
    from_paf = mp.tokenise_paf_device(open(str(paf), "rb"), acc_index)
    host_paf = mp.tokenise_paf(paf_lines, acc_index)
    assert len(from_sam) == len(sam_lines)
    assert np.array_equal(from_paf, from_sam) and np.array_equal(host_paf, from_sam)
    # the whole stage through map_main: .paf (replayed like a SAM file, all the way on the device) == .sam, byte for byte
    outs = []
    for infile in (sam, paf):
        out = tmp_path / (infile.name + ".tsv")
        args = sc.make_args(str(infile), str(dbinfo), str(out), {"input_type": "AUTO", "sampleID": "s"})
        mp.map_main(args)
        outs.append(out.read_text())
    assert outs[0] == outs[1] and outs[0].count("\n") > 20
    # ... and with the records staying on the device end to end (--device_multimap)
    out = tmp_path / "dm.tsv"
    args = sc.make_args(str(paf), str(dbinfo), str(out), {"input_type": "AUTO", "sampleID": "s", "device_multimap": True})
    mp.map_main(args)
    a = [ln.split("\t") for ln in outs[0].splitlines() if ln and ln[0] != "@"]
    b = [ln.split("\t") for ln in out.read_text().splitlines() if ln and ln[0] != "@"]
    assert [r[0] for r in a] == [r[0] for r in b]
    assert all(abs(float(x[4]) - float(y[4])) <= 1e-6 for x, y in zip(a, b))


def test_paf_device_tokeniser_edges(hip):
    acc_index = {"Unmapped": 0, "NZ_A.1": 1, "NZ_B.1": 2}
    good = ("r1\t40\t5\t40\t+\tNZ_A.1\t5000\t9\t44\t34\t35\t60\ttp:A:P\tcg:Z:35M\n"
            "short\tline\twith\tfew\tfields\n"
            "r1\t40\t0\t38\t-\tNZ_B.1\t5000\t98\t138\t36\t40\t0\ttp:A:P\ttp:A:S\tcg:Z:1M\tcg:Z:30M2D8M\r\n"
            "r2\t100\t0\t100\t+\tNZ_A.1\t5000\t0\t100\t90\t100\t60\n"           # exactly 12 fields, no tags
            "r3\t100\t0\t100\t+\tNZ_A.1\t5000\t0\t100\t90\t100\t60\ttp:A:P")     # no trailing newline
    want = mp.tokenise_paf(good.splitlines(True), acc_index)
    got = mp.tokenise_paf_device([good.encode()], acc_index)
    assert len(want) == 4 and np.array_equal(got, want)
    assert (int(want["matched"][1]), int(want["total"][1]), int(want["flag_len"][1])) == (38, 42, 16 | 256)  # the LAST tags win
    for bad, exc in (("r1\t40\t5\t40\t+\tNOPE\t5000\t9\t44\t34\t35\t60\ttp:A:P\n", KeyError),
                     ("r1\tx40\t5\t40\t+\tNZ_A.1\t5000\t9\t44\t34\t35\t60\ttp:A:P\n", ValueError),
                     ("r1\t0\t0\t0\t+\tNZ_A.1\t5000\t9\t44\t0\t35\t60\n", ZeroDivisionError)):
        with pytest.raises(exc):
            mp.tokenise_paf(bad.splitlines(True), acc_index)
        with pytest.raises(exc):
            mp.tokenise_paf_device([good.encode() + b"\n" + bad.encode()], acc_index)


def test_lines_the_host_accepts_but_the_device_parser_does_not_go_through_the_host(hip):
    """Python's int() takes surrounding blanks, '_' between digits and numbers of any size; the kernel's integer
    parser takes digits.  The host tokeniser is the definition: a chunk with such a line is tokenised on the host,
    with the previous QNAME carried in and out, instead of failing (the bare re-raise)."""
    acc_index = {"Unmapped": 0, "NZ_A.1": 1, "NZ_B.1": 2}
    text = ("r1\t40\t5\t40\t+\tNZ_A.1\t5000\t9\t44\t34\t35\t60\ttp:A:P\tcg:Z:35M\n"
            "r1\t 40 \t0\t3_8\t-\tNZ_B.1\t5000\t98\t138\t36\t40\t0\ttp:A:S\tcg:Z:30M2D8M\n"   # ' 40 ' and '3_8': int() takes both
            "r2\t100\t0\t100\t+\tNZ_A.1\t5000\t0\t100\t90\t100\t60\n"
            "r2\t100\t0\t100\t+\tNZ_B.1\t5000\t0\t100\t0090\t100\t60\ttp:A:S\n")
    want = mp.tokenise_paf(text.splitlines(True), acc_index)
    got = mp.tokenise_paf_device([text.encode()], acc_index)
    assert len(want) == 4 and np.array_equal(got, want)
    # two chunks: the second starts in the middle of read r2 — the carried QNAME keeps its new-read bit clear
    lines = text.splitlines(True)
    got2 = mp.tokenise_paf_device(iter([l.encode() for l in lines]), acc_index)  # (an iterator of lines is batched into one chunk)
    assert np.array_equal(got2, want)
    old = mp._CHUNK_BYTES
    mp._CHUNK_BYTES = 64
    try:
        got3 = mp.tokenise_paf_device(iter([l.encode() for l in lines]), acc_index)
    finally:
        mp._CHUNK_BYTES = old
    assert np.array_equal(got3, want)
