"""Stage A by k-mer identity, per-lane code compiled for the host (tests/host_kcount_check.cpp over
metalign_amd/csrc/mg_kcount_core.h) against the oracle's mgo_refpipe_count_kmers: the packing of the base stream, the sliding
minimizer in all three modes, restarts after a full event list, the table-side minimizer, gate, buckets, signatures and the
exact comparison — for k from 15 to 64, reads with N runs and lower case, ragged and empty reads, reads in chunks (longer than
1023), tiles that start off a 16-byte boundary.  (The GPU tests hold the kernel itself to the same oracle; this runs where
there is no GPU.)"""
import os
import subprocess

import numpy as np
import pytest

import oracle

HERE = os.path.dirname(os.path.abspath(__file__))
ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = bytes.maketrans(b"ACGT", b"TGCA")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("kcount") / "host_kcount_check")
    subprocess.check_call(["g++", "-O1", "-std=c++20", "-o", exe, os.path.join(HERE, "host_kcount_check.cpp")])
    return exe


def unpack(hi, lo, k):
    v = (int(hi) << 64) | int(lo)
    return bytes(b"ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def workload(rng, k, kind):
    """-> (table k-mers as strings, pair-order kmer_hi / kmer_lo, reads as bytes)"""
    ngen, glen, n = 6, 3000, 60
    genomes = [rng.choice(ALPHA, size=glen).astype(np.uint8) for _ in range(ngen)]
    genomes[1][500:520] = ord("A")                       # a homopolymer: every m-mer of it is one value
    genomes[2][100:160] = np.tile(np.frombuffer(b"ACGTTGCA", dtype=np.uint8), 8)[:60]  # a tandem repeat
    genomes[3] = np.frombuffer(bytes(genomes[0]).translate(COMP)[::-1], dtype=np.uint8).copy()  # a reverse-complemented genome
    gb = np.concatenate(genomes)
    go = (np.arange(ngen + 1) * glen).astype(np.uint64)
    h, khi, klo, o = oracle.sketch_genomes_kmers(gb, go, k, n)
    # a few entries from the special stretches, whatever their hash, in the orientation they stand in (and one twice)
    extra = [bytes(genomes[1][495:495 + k]), bytes(genomes[2][90:90 + k]), bytes(genomes[2][98:98 + k]), bytes(genomes[1][495:495 + k])]
    table = [unpack(a, b, k) for a, b in zip(khi, klo)] + extra
    reads = []
    for _ in range(200 if kind != "long" else 70):
        g = int(rng.integers(0, ngen))
        if kind == "equal":
            L = 150
        elif kind == "long":
            L = int(rng.choice([1500, 2600, 40, 1023, 1024]))
        else:
            L = int(rng.choice([0, 1, 14, k - 1, k, k + 1, 100, 150, 151, 250]))
        L = min(L, glen)
        st = int(rng.integers(0, glen - L + 1))
        r = bytearray(genomes[g][st:st + L])
        if rng.random() < 0.5:
            r = bytearray(bytes(r).translate(COMP)[::-1])
        for j in range(len(r)):
            if rng.random() < 0.01:
                r[j] = int(rng.choice(ALPHA))
        if kind in ("ragged", "long"):
            if rng.random() < 0.3 and len(r) > 5:
                j = int(rng.integers(0, len(r) - 3))
                r[j:j + int(rng.integers(1, 4))] = b"N" * 1
            if rng.random() < 0.2:
                r = bytearray(bytes(r).lower())
        reads.append(bytes(r))
    # reads that ARE the special stretches, several times (ties between equal minimizers; counts above one)
    for _ in range(3):
        reads.append(bytes(genomes[1][470:470 + 120]))
        reads.append(bytes(genomes[2][60:60 + 140]))
    if kind != "equal":
        reads.append(b"")
    else:
        reads = [r for r in reads if len(r) == 150 or len(r) == 120 or len(r) == 140]
        reads = [r for r in reads if len(r) == 150]
    return table, reads


def pack_table(table, k):
    hi = np.zeros(len(table), dtype=np.uint64)
    lo = np.zeros(len(table), dtype=np.uint64)
    for i, t in enumerate(table):
        v = 0
        for ch in t:
            v = (v << 2) | b"ACGT".index(ch)
        hi[i], lo[i] = v >> 64, v & 0xFFFFFFFFFFFFFFFF
    return hi, lo


@pytest.mark.parametrize("k", [15, 16, 17, 21, 31, 32, 33, 47, 51, 60, 63, 64])
@pytest.mark.parametrize("kind,cap,lead,cs,bb", [("equal", 12, 0, 0, 0), ("ragged", 12, 5, 2, 0), ("ragged", 3, 0, 0, 0), ("long", 12, 3, 3, 0),
                                                 ("equal", 3, 9, 1, 0), ("ragged", 12, 2, 2, 4)])
def test_counts_equal_the_oracle(checker, k, kind, cap, lead, cs, bb):
    oracle.build()
    rng = np.random.default_rng(1000 * k + cap + lead)
    table, reads = workload(rng, k, kind)
    text = ("%d %d %d %d %d %d %d\n" % (k, cap, len(table), len(reads), lead, cs, bb)).encode() + b"".join(t + b"\n" for t in table) + \
        b"".join(r + b"\n" for r in reads)
    out = subprocess.run([checker], input=text, capture_output=True, check=True).stdout.decode().split("\n")
    got = np.array([int(x) for x in out[:len(table)]], dtype=np.uint32)
    tail = out[len(table)].split()
    bases = np.frombuffer(b"".join(reads), dtype=np.uint8)
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    khi, klo = pack_table(table, k)
    want, seen = oracle.refpipe_count_kmers(bases, offs, k, khi, klo, cs=cs)
    assert int(tail[1]) == seen, "k-mers of the reads"
    assert np.array_equal(got, want), "counts differ at %s" % np.flatnonzero(got != want)[:10]
    assert want.sum() > 0
    if cap == 3:
        assert int(tail[7]) > 0, "a list of three slots must have forced restarts"
