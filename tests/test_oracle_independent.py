"""CPU suite: the C oracle (oracle/mg_oracle.c) against the independent pure-Python restatement (tests/indep_sketch.py).

Stage A/B parity is "unpinned" (KMC 3 / CMash are absent from the reference tree): nothing here makes
it green.  What this removes is single-implementation risk: the two restatements were written from the same prose
 in different styles and must agree bit for bit, MurmurHash3 included.
"""
import json
import os

import numpy as np
import pytest

import indep_sketch as ind

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(params=[0, 1], ids=["canonical_kmer_hash", "cmash_recollection"])
def hash_mode(request, oracle_lib):
    """Both definitions of a k-mer's hash: 0 = hash(min(kmer, revcomp)), 64 bits — the default; 1 =
    min(hash(kmer), hash(revcomp)) % 9999999999971, CMash as SURVEY.md §8(c) recollects it (unverified)."""
    oracle_lib.set_hash_mode(request.param)
    ind.HASH_MODE = request.param
    yield request.param
    oracle_lib.set_hash_mode(0)
    ind.HASH_MODE = 0


def test_independent_murmur3_matches_the_known_answers():
    with open(os.path.join(HERE, "golden", "murmur3_kat.json")) as fh:
        kat = json.load(fh)
    assert len(kat) >= 56
    for v in kat:
        assert ind.murmur3_x64_128(bytes.fromhex(v["hex"]), v["seed"]) == (v["h1"], v["h2"]), v["hex"]
    # the one value published in mmh3's README: mmh3.hash64("foo") == (-2129773440516405919, 9128664383759220103)
    h1, h2 = ind.murmur3_x64_128(b"foo", 0)
    assert (h1 - (1 << 64), h2) == (-2129773440516405919, 9128664383759220103)


def _reads(rng, n, lo, hi, p_n=0.01, p_lower=0.1):
    out = []
    for _ in range(n):
        ln = int(rng.integers(lo, hi + 1))
        b = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=ln).astype(np.uint8)
        b[rng.random(ln) < p_n] = ord("N")
        m = rng.random(ln) < p_lower
        b[m] |= 0x20
        out.append(b.tobytes())
    return out


def _flat(reads):
    offs = np.zeros(len(reads) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in reads])
    return np.frombuffer(b"".join(reads), dtype=np.uint8), offs


@pytest.mark.parametrize("k", [1, 4, 21, 32, 33, 51, 60, 64])
@pytest.mark.parametrize("cs", [0, 3])
def test_read_sketch_c_oracle_equals_independent(oracle_lib, hash_mode, k, cs):
    rng = np.random.default_rng(1000 + k)
    # ragged reads incl. shorter than k and empty, N and lower case; a repeated read so that counts exceed cs
    reads = _reads(rng, 60, 0, 150) + [b"", b"ACGT", b"N" * 70]
    reads += [reads[3]] * 5
    bases, offs = _flat(reads)
    top = ind.CMASH_PRIME if hash_mode else 2 ** 64
    for hmax, s in ((ind.M64, 0), (int(0.3 * top), 0), (ind.M64, 37)):
        oh, oc, otr, oseen = oracle_lib.sketch_reads(bases, offs, k, hmax=hmax, s=s, cs=cs)
        items, tr, seen = ind.sketch_reads(reads, k, hmax=hmax, s=s, cs=cs)
        assert [int(x) for x in oh] == [h for h, _ in items]
        assert [int(x) for x in oc] == [c for _, c in items]
        assert (otr, oseen) == (tr, seen)


def test_strand_and_case_invariance_of_both(oracle_lib, hash_mode):
    rng = np.random.default_rng(5)
    reads = _reads(rng, 20, 80, 120, p_n=0.0, p_lower=0.0)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    flipped = [r.translate(comp)[::-1].lower() for r in reads]
    a, _, _ = ind.sketch_reads(reads, 31)
    b, _, _ = ind.sketch_reads(flipped, 31)
    assert a == b
    bb, bo = _flat(flipped)
    oh, oc, _, _ = oracle_lib.sketch_reads(bb, bo, 31, cs=0)
    assert [int(x) for x in oh] == [h for h, _ in a] and [int(x) for x in oc] == [c for _, c in a]


@pytest.mark.parametrize("k", [21, 33, 60])
def test_genome_sketch_and_containment_c_oracle_equals_independent(oracle_lib, hash_mode, k):
    rng = np.random.default_rng(77 + k)
    genomes = _reads(rng, 6, 1500, 2500, p_n=0.002, p_lower=0.05) + [b"ACGT" * 3]  # the last one shorter than most k
    n = 120
    gb, go = _flat(genomes)
    dbh, dbo = oracle_lib.sketch_genomes(gb, go, k, n)
    want = [ind.sketch_genome(g, k, n) for g in genomes]
    for g, w in enumerate(want):
        assert [int(x) for x in dbh[int(dbo[g]):int(dbo[g + 1])]] == w
    # reads from genomes 1 and 4, twice each so that ci = 2 is met, plus noise
    reads = []
    for g in (1, 4):
        src = genomes[g]
        for _ in range(40):
            a = int(rng.integers(0, len(src) - 150))
            reads += [src[a:a + 150]] * 2
    reads += _reads(rng, 30, 100, 150)
    rb, ro = _flat(reads)
    hmax = int(dbh.max())
    for s in (0, 150):
        oh, oc, otr, _ = oracle_lib.sketch_reads(rb, ro, k, hmax=hmax, s=s)
        items, tr, _ = ind.sketch_reads(reads, k, hmax=hmax, s=s, cs=oracle_lib.DEFAULT_CS)
        assert [int(x) for x in oh] == [h for h, _ in items] and otr == tr
        hits, sizes = oracle_lib.containment(oh, oc, otr, 2, dbh, dbo)
        assert [(int(h), int(z)) for h, z in zip(hits, sizes)] == ind.containment(items, tr, 2, want)
    assert hits[1] > hits[0] and hits[4] > hits[0]


def test_filtered_sketch_c_oracle_equals_independent(oracle_lib):
    rng = np.random.default_rng(9)
    k = 21
    genomes = _reads(rng, 4, 3000, 3000, p_n=0.0, p_lower=0.0)
    gb, go = _flat(genomes)
    dbh, _ = oracle_lib.sketch_genomes(gb, go, k, 200)
    reads = [genomes[2][i:i + 100] for i in range(0, 2900, 7)] + _reads(rng, 50, 100, 100)
    rb, ro = _flat(reads)
    hmax = int(dbh.max())
    oh, oc, otr, _ = oracle_lib.sketch_reads_filtered(rb, ro, k, dbh, hmax=hmax)
    bits, mask = oracle_lib.filter_bits(dbh)
    items, tr, _ = ind.sketch_reads(reads, k, hmax=hmax, cs=oracle_lib.DEFAULT_CS, member=lambda h: bool(bits[h & int(mask)]))
    assert [int(x) for x in oh] == [h for h, _ in items] and [int(x) for x in oc] == [c for _, c in items]


def test_the_cmash_recollection_is_what_its_words_say(oracle_lib):
    """Mode 1 spelled out on one k-mer: both strands hashed, the smaller VALUE kept, modulo the prime — and the two modes
    differ (the smaller strand's hash is not the smaller hash)."""
    kmer, rc = b"ACGTTGCAAGGCTTAAACCCG", b"CGGGTTTAAGCCTTGCAACGT"
    a, b = ind.murmur3_x64_128(kmer)[0], ind.murmur3_x64_128(rc)[0]
    assert ind.CMASH_PRIME == 9999999999971 and all(ind.CMASH_PRIME % p for p in range(2, 10000))
    oracle_lib.set_hash_mode(1)
    try:
        h, v = oracle_lib.kmer_hashes(kmer, 21)
        assert v[0] and int(h[0]) == min(a, b) % ind.CMASH_PRIME
        h2, _ = oracle_lib.kmer_hashes(rc.lower(), 21)
        assert int(h2[0]) == int(h[0])
    finally:
        oracle_lib.set_hash_mode(0)
    h0, _ = oracle_lib.kmer_hashes(kmer, 21)
    assert int(h0[0]) == a and a != min(a, b) % ind.CMASH_PRIME  # (ACGTT... < CGGGT...: mode 0 hashes the forward strand)


@pytest.mark.parametrize("kmax,k", [(60, 30), (51, 21), (33, 32), (21, 21), (9, 4)])
def test_prefix_tables_of_the_cmash_recollection(oracle_lib, kmax, k):
    """The k < k_max tables of hash mode 1 (k-prefixes of the sketched k_max-mers): C oracle == independent restatement;
    with k == k_max the table is the mode-1 sketch itself."""
    rng = np.random.default_rng(31 * kmax + k)
    genomes = _reads(rng, 5, 900, 1600, p_n=0.003, p_lower=0.05) + [b"ACGT" * 4, b"A" * 200]
    n = 90
    gb, go = _flat(genomes)
    h, o = oracle_lib.sketch_genomes_prefix(gb, go, kmax, k, n)
    for g, seq in enumerate(genomes):
        assert [int(x) for x in h[int(o[g]):int(o[g + 1])]] == ind.sketch_genome_prefix(seq, kmax, k, n), g
    if k == kmax:
        oracle_lib.set_hash_mode(1)
        try:
            dbh, dbo = oracle_lib.sketch_genomes(gb, go, kmax, n)
        finally:
            oracle_lib.set_hash_mode(0)
        assert np.array_equal(h, dbh) and np.array_equal(o, dbo)


# ---- the reference's own wiring of stage A/B: k_max-mers only on the read side, prefix columns from the table side -------------
def _unpack(hi, lo, k):
    v = (int(hi) << 64) | int(lo)
    return bytes(b"ACGT"[(v >> (2 * (k - 1 - i))) & 3] for i in range(k))


def _refpipe_case(rng, strains=True):
    """Genomes that SHARE k-mers (a strain = a mutated copy; a repeat inside one genome; a reverse-complemented copy), N runs,
    lower case; reads from three of them, both strands, each twice so that ci = 2 is met, plus reads of nothing."""
    genomes = _reads(rng, 5, 1200, 2000, p_n=0.002, p_lower=0.05)
    if strains:
        g0 = bytearray(genomes[0].upper())
        for p in rng.integers(0, len(g0), size=12):
            g0[int(p)] = b"ACGT"[int(rng.integers(0, 4))]
        genomes.append(bytes(g0))                                                   # a strain of genome 0
        genomes.append(genomes[1][:600] + genomes[1][100:700])                      # repeats inside one genome
        genomes.append(genomes[2].upper().translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1])  # the other strand of genome 2
    genomes += [b"ACGT" * 5, b"", b"A" * 300]
    reads = []
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    for g in (0, 2, 4):
        src = genomes[g]
        for _ in range(60):
            a = int(rng.integers(0, len(src) - 150))
            r = src[a:a + 150]
            if rng.random() < 0.5:
                r = r.translate(comp)[::-1]
            reads += [r, r]
    reads += _reads(rng, 40, 60, 150) + [b"", b"ACG"]
    return genomes, reads


@pytest.mark.parametrize("ks", [[21, 31, 51], [30, 40, 50, 60], [4, 6, 9], [5, 33, 64], [32], [8, 16, 32]], ids=str)
def test_reference_pipeline_c_oracle_equals_the_string_restatement(oracle_lib, hash_mode, ks):
    rng = np.random.default_rng(4000 + 7 * sum(ks) + hash_mode)
    genomes, reads = _refpipe_case(rng)
    kmax, n = ks[-1], 150
    gb, go = _flat(genomes)
    h, khi, klo, o = oracle_lib.sketch_genomes_kmers(gb, go, kmax, n)
    # the table's k-mers: the same hashes as the plain sketch, the kept k-mers as the string restatement keeps them
    ph, po = oracle_lib.sketch_genomes(gb, go, kmax, n)
    assert np.array_equal(h, ph) and np.array_equal(o, po)
    want_kmers = [ind.refpipe_genome_kmers(g, kmax, n) for g in genomes]
    for g, w in enumerate(want_kmers):
        got = [_unpack(khi[e], klo[e], kmax) for e in range(int(o[g]), int(o[g + 1]))]
        assert got == w, g
    table = oracle_lib.refpipe_build(h, khi, klo, o, ks)
    assert np.all(np.diff(table["pair_hash"].astype(object)) >= 0)
    for k in ks[:-1]:
        t = table["small"][k]
        d = sorted({y[:k] for g in want_kmers for y in g})
        assert t["nprefix"] == len(d)
        # pa / pb name the prefixes they should (checked through the kept k-mers in pair order)
        for i in rng.integers(0, len(table["pair_hash"]), size=min(200, len(table["pair_hash"]))):
            y = _unpack(table["kmer_hi"][i], table["kmer_lo"][i], kmax)
            assert d[int(t["pa"][i])] == y[:k]
            other = ind._revcomp(y)[:k]
            assert (d[int(t["pb"][i])] == other) if t["pb"][i] != 0xFFFFFFFF else (other not in set(d))
        assert [int(x) for x in t["gsize"]] == [len({y[:k] for y in g}) for g in want_kmers]
    rb, ro = _flat(reads)
    for ci in (1, 2):
        qh, qc, _, _ = oracle_lib.sketch_reads(rb, ro, kmax, hmax=int(h.max()) if len(h) else 0)
        hits, sizes = oracle_lib.refpipe_containment(qh, qc, ci, table)
        want = ind.refpipe_query(reads, want_kmers, ks, ci)
        for ki in range(len(ks)):
            assert [(int(a), int(b)) for a, b in zip(hits[ki], sizes[ki])] == want[ki], (ks[ki], ci)
        # the largest k's column is the plain containment of the k_max table
        ph_hits, ph_sizes = oracle_lib.containment(qh, qc, False, ci, h, o)
        assert np.array_equal(hits[-1], ph_hits) and np.array_equal(sizes[-1], ph_sizes)
        # stage A BY K-MER IDENTITY (mgo_refpipe_count_kmers: what kmc + kmc_tools intersect compute, no hash on the read side):
        # the same columns — and, per sketched k-mer, the string restatement's own count of the reads' canonical k-mers
        counts, seen = oracle_lib.refpipe_count_kmers(rb, ro, kmax, table["kmer_hi"], table["kmer_lo"], cs=0)
        khits, ksizes = oracle_lib.refpipe_containment_counts(counts, ci, table)
        assert np.array_equal(khits, hits) and np.array_equal(ksizes, sizes)
        if ci == 1:
            occ = {}
            for r in reads:
                for m in ind._RUNS.finditer(r):
                    run = m.group().upper()
                    for i in range(len(run) - kmax + 1):
                        x = run[i:i + kmax]
                        c = min(x, ind._revcomp(x))
                        occ[c] = occ.get(c, 0) + 1
            assert seen == sum(occ.values())
            for i in rng.integers(0, len(counts), size=min(300, len(counts))):
                y = _unpack(table["kmer_hi"][i], table["kmer_lo"][i], kmax)
                assert int(counts[i]) == occ.get(min(y, ind._revcomp(y)), 0)
            sat = oracle_lib.refpipe_count_kmers(rb, ro, kmax, table["kmer_hi"], table["kmer_lo"], cs=3)[0]
            assert np.array_equal(sat, np.minimum(counts, 3))
    # genomes that were sampled stand out at every k; a k-prefix column is never below the k_max column's hits / never above its size
    for ki in range(len(ks)):
        assert hits[ki][0] > 0 and hits[ki][2] > 0 and hits[ki][4] > 0
        assert np.all(hits[ki] <= sizes[ki])


@pytest.mark.parametrize("ks", [[21, 31, 51], [30, 40, 50, 60], [4, 6, 10], [32]], ids=str)
def test_forward_selected_sketches_c_oracle_equals_the_string_restatement(oracle_lib, hash_mode, ks):
    """`build_db --sketch_hash forward`: the entries are chosen by the hash of the k-mer AS IT STANDS and kept as they stand; the query
    side is unchanged.  One genome holds a stretch and its reverse complement, so a k-mer and its reverse complement are both in its
    sketch: two strings, one matching identity, both counted."""
    rng = np.random.default_rng(4400 + 7 * sum(ks) + hash_mode)
    genomes, reads = _refpipe_case(rng)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    genomes[3] = genomes[3].upper()[:500] + genomes[3].upper()[100:400].translate(comp)[::-1]
    src = genomes[3]
    for a in range(0, len(src) - 150, 40):
        reads += [src[a:a + 150]] * 2
    kmax, n = ks[-1], 150
    gb, go = _flat(genomes)
    h, khi, klo, o = oracle_lib.sketch_genomes_kmers(gb, go, kmax, n, sketch_hash="forward")
    want_kmers = [ind.refpipe_genome_kmers_forward(g, kmax, n) for g in genomes]
    twice = 0
    for g, w in enumerate(want_kmers):
        got = [_unpack(khi[e], klo[e], kmax) for e in range(int(o[g]), int(o[g + 1]))]
        assert got == w, g
        ids = [int(x) for x in h[int(o[g]):int(o[g + 1])]]
        twice += len(ids) - len(set(ids))
        # what an entry matches by: the hash of its k-mer under the mode in force, as the canonical sketch computes it
        for y, v in list(zip(w, ids))[:40]:
            hf, hr = ind.murmur3_x64_128(y, 0)[0], ind.murmur3_x64_128(ind._revcomp(y), 0)[0]
            assert v == (min(hf, hr) % ind.CMASH_PRIME if hash_mode == 1 else ind.murmur3_x64_128(min(y, ind._revcomp(y)), 0)[0])
    if kmax >= 21:
        assert twice > 0  # (genome 3 lists a k-mer and its reverse complement)
    table = oracle_lib.refpipe_build(h, khi, klo, o, ks)
    for k in ks[:-1]:
        t = table["small"][k]
        assert t["nprefix"] == len({y[:k] for g in want_kmers for y in g})
        assert [int(x) for x in t["gsize"]] == [len({y[:k] for y in g}) for g in want_kmers]
    rb, ro = _flat(reads)
    for ci in (1, 2):
        qh, qc, _, _ = oracle_lib.sketch_reads(rb, ro, kmax, hmax=int(h.max()) if len(h) else 0)
        hits, sizes = oracle_lib.refpipe_containment(qh, qc, ci, table)
        want = ind.refpipe_query(reads, want_kmers, ks, ci)
        for ki in range(len(ks)):
            assert [(int(a), int(b)) for a, b in zip(hits[ki], sizes[ki])] == want[ki], (ks[ki], ci)
    assert hits[-1][3] > 0


def test_reference_pipeline_smaller_k_columns_are_not_independent_sketches(oracle_lib):
    """What separates the reference's wiring from a sketch per k: a read that holds a genome's 21-mers but none of its 51-mers
    (every 51-mer is broken by an error) contributes NOTHING to the k = 21 column."""
    rng = np.random.default_rng(5)
    genome = _reads(rng, 1, 3000, 3000, p_n=0.0, p_lower=0.0)[0]
    ks, n = [21, 51], 400
    gb, go = _flat([genome])
    h, khi, klo, o = oracle_lib.sketch_genomes_kmers(gb, go, 51, n)
    table = oracle_lib.refpipe_build(h, khi, klo, o, ks)
    reads = []
    for a in range(0, 2800, 50):  # 100-base reads with a substitution every 40 bases: 21-mers survive, 51-mers do not
        r = bytearray(genome[a:a + 100])
        for p in range(20, 100, 40):
            r[p] = ord("A") if r[p] != ord("A") else ord("C")
        reads += [bytes(r)] * 2
    rb, ro = _flat(reads)
    qh, qc, _, _ = oracle_lib.sketch_reads(rb, ro, 51, hmax=int(h.max()))
    hits, _ = oracle_lib.refpipe_containment(qh, qc, 2, table)
    assert hits[1][0] == 0 and hits[0][0] == 0
    want = ind.refpipe_query(reads, [ind.refpipe_genome_kmers(genome, 51, n)], ks, 2)
    assert want[0][0][0] == 0 and want[1][0][0] == 0
    # ... while an independent 21-mer sketch of the same reads does find the genome
    g21, _ = oracle_lib.sketch_genomes(gb, go, 21, n)
    q21, c21, _, _ = oracle_lib.sketch_reads(rb, ro, 21, hmax=int(g21.max()))
    hh, _ = oracle_lib.containment(q21, c21, False, 2, g21, np.asarray([0, len(g21)], dtype=np.uint64))
    assert hh[0] > 50
