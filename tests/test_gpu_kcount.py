"""GPU parity of stage A BY K-MER IDENTITY (mg_kcount.hip: the minimizer-partitioned count of the reads' k_max-mers against the
table's k-mers, what `kmc` + `kmc_tools intersect` compute, scripts/select_db.py:50-59) against the oracle's
mgo_refpipe_count_kmers, through the C ABI: per-pair counts, KMC's total of k-mers, the columns of every k, and the hash path
(mg_sketch_* + mg_refpipe_containment_dev) on the same inputs.  Bit-exact: integer work throughout."""
import numpy as np
import pytest

from util import flat, random_genomes, refpipe_case, sample_reads

pytestmark = pytest.mark.gpu

K_SETS = [[21, 31, 51], [30, 40, 50, 60], [15], [16, 17], [5, 33, 64], [32], [20, 47], [31, 32, 33, 63]]


def _counts(hip, table, reads, pieces=1):
    kc = table.kmer_counts()
    step = (len(reads) + pieces - 1) // pieces
    keep = []
    for a in range(0, max(len(reads), 1), max(step, 1)):
        rb, ro = flat(reads[a:a + step])
        d_b, d_o = hip.array(np.concatenate([rb, np.zeros(64, np.uint8)])), hip.array(ro)
        kc.add_dev(d_b.ptr, d_o.ptr, len(ro) - 1, int(ro[-1]))
        keep += [d_b, d_o]
    hip.sync()
    for x in keep:
        x.free()
    return kc


# (tables selected by the forward hash are built for a list of k: include/metalign_hip.h, mg_sketch_genomes_kmers_forward)
CASES = [(ks, "canonical") for ks in K_SETS] + [(ks, "forward") for ks in ([21, 31, 51], [30, 40, 50, 60], [15], [5, 33, 64], [32])]


@pytest.mark.parametrize("ks,sketch_hash", CASES, ids=str)
def test_counts_and_columns_match_the_oracle(hip, oracle_lib, ks, sketch_hash):
    rng = np.random.default_rng(6100 + 17 * sum(ks) + len(sketch_hash))
    genomes, reads = refpipe_case(rng)
    kmax, n = ks[-1], 150
    gb, go = flat(genomes)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, kmax, n, sketch_hash=sketch_hash)
    table = hip.refdb_build(h, khi, klo, o, ks)
    want_table = oracle_lib.refpipe_build(h, khi, klo, o, ks)
    table.index_kmers()
    assert table.has_kmer_index and 0 < table.distinct_kmers <= len(h)
    rb, ro = flat(reads)
    for cs in (3, 0, 1):
        hip.count_saturation(cs)
        try:
            want, seen = oracle_lib.refpipe_count_kmers(rb, ro, kmax, want_table["kmer_hi"], want_table["kmer_lo"], cs=cs)
            kc = _counts(hip, table, reads, pieces=3 if cs == 3 else 1)
            got = kc.download()
            st = kc.stats()
            assert np.array_equal(got, want), np.flatnonzero(got != want)[:10]
            assert st["kmers"] == seen
            assert want.max() >= (cs if cs else 2)
            for ci in (1, 2, 3):
                if cs and ci > cs:
                    continue
                hits, sizes = hip.refpipe_containment_counts(kc, table, ci)
                whits, wsizes = oracle_lib.refpipe_containment_counts(want, ci, want_table)
                assert np.array_equal(hits, whits), (ks, cs, ci)
                assert np.array_equal(sizes, wsizes)
            kc.free()
        finally:
            hip.count_saturation(3)
    # the hash path on the same inputs: the same columns (no two k-mers of this table share a hash)
    d_b, d_o = hip.array(np.concatenate([rb, np.zeros(64, np.uint8)])), hip.array(ro)
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(reads), kmax, table.max_hash, 0)
    hits_h, sizes_h = hip.refpipe_containment(sk, table, 2)
    kc = _counts(hip, table, reads)
    hits_k, sizes_k = hip.refpipe_containment_counts(kc, table, 2)
    assert np.array_equal(hits_h, hits_k) and np.array_equal(sizes_h, sizes_k)
    # heads: a pair counts at the first pair with its k-mer
    heads = table.kmer_heads()
    assert np.all(heads <= np.arange(len(heads))) and np.array_equal(heads[heads], heads)
    for x in (sk, kc, table, d_b, d_o):
        x.free()


def test_an_uploaded_table_is_indexed_from_its_stored_kmers(hip, oracle_lib):
    ks = [21, 31, 51]
    rng = np.random.default_rng(6200)
    genomes, reads = refpipe_case(rng)
    gb, go = flat(genomes)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], 150)
    built = hip.refdb_build(h, khi, klo, o, ks)
    got = built.download()
    up = hip.refdb_upload(ks, len(genomes), got["pair_hash"], got["pair_gen"], got["gsize"], built.max_hash, [got["small"][k] for k in ks[:-1]])
    with pytest.raises(Exception):
        up.index_kmers()  # an uploaded table does not hold its k-mers
    up.index_kmers(got["kmer_hi"], got["kmer_lo"])
    built.index_kmers()
    a, b = _counts(hip, up, reads), _counts(hip, built, reads)
    assert np.array_equal(a.download(), b.download())
    ha, _ = hip.refpipe_containment_counts(a, up, 2)
    hb, _ = hip.refpipe_containment_counts(b, built, 2)
    assert np.array_equal(ha, hb) and ha.sum() > 0
    for x in (a, b, up, built):
        x.free()


@pytest.mark.parametrize("kind", ["long", "tiny_stage", "one_read", "none", "all_n"])
def test_reads_of_every_shape(hip, oracle_lib, kind):
    """Reads longer than 1023 (taken through in chunks), a batch whose average length sizes a stage the longest read does not
    fit, a single read, no read at all, reads of N only."""
    ks, n = [31, 51], 200
    rng = np.random.default_rng(6300 + len(kind))
    gb, go = random_genomes(rng, 6, 6000)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], n)
    table = hip.refdb_build(h, khi, klo, o, ks)
    want_table = oracle_lib.refpipe_build(h, khi, klo, o, ks)
    table.index_kmers()
    g = [bytes(gb[int(go[i]):int(go[i + 1])]) for i in range(6)]
    if kind == "long":
        reads = [g[0][:2500], g[1][100:1400], g[2][:1023], g[2][:1024], g[3], g[0][:2500]] + [g[4][i:i + 150] for i in range(0, 3000, 50)]
    elif kind == "tiny_stage":
        reads = [g[0][i:i + 60] for i in range(0, 5000, 7)] + [g[1][:900]] + [g[0][i:i + 60] for i in range(0, 5000, 7)]
    elif kind == "one_read":
        reads = [g[5][200:350]]
    elif kind == "none":
        reads = []
    else:
        reads = [b"N" * 150] * 70 + [g[0][:150], g[0][:150]]
    rb, ro = flat(reads)
    want, seen = oracle_lib.refpipe_count_kmers(rb, ro, ks[-1], want_table["kmer_hi"], want_table["kmer_lo"], cs=3)
    kc = _counts(hip, table, reads) if reads else table.kmer_counts()
    got = kc.download()
    assert np.array_equal(got, want)
    assert kc.stats()["kmers"] == seen
    if kind in ("long", "tiny_stage"):
        assert want.sum() > 0
    kc.reset()
    hip.sync()
    assert kc.download().sum() == 0 and kc.stats()["kmers"] == 0
    kc.free()
    table.free()


def test_a_larger_sample_against_a_larger_table(hip, oracle_lib):
    """40 genomes x 500 sketched k-mers, 30 000 reads of 150 bp with errors from a few of them, both strands: every pair's count."""
    ks = [30, 40, 50, 60]
    rng = np.random.default_rng(6400)
    gb, go = random_genomes(rng, 40, 20000)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], 500)
    table = hip.refdb_build(h, khi, klo, o, ks)
    want_table = oracle_lib.refpipe_build(h, khi, klo, o, ks)
    table.index_kmers()
    rb, ro, _ = sample_reads(rng, gb, go, 30000, 150, err=0.01, present=[1, 5, 7, 30])
    reads = [bytes(rb[int(ro[i]):int(ro[i + 1])]) for i in range(len(ro) - 1)]
    want, seen = oracle_lib.refpipe_count_kmers(rb, ro, ks[-1], want_table["kmer_hi"], want_table["kmer_lo"], cs=3)
    kc = _counts(hip, table, reads, pieces=2)
    assert np.array_equal(kc.download(), want)
    st = kc.stats()
    assert st["kmers"] == seen and st["matches"] > 0
    hits, sizes = hip.refpipe_containment_counts(kc, table, 2)
    whits, wsizes = oracle_lib.refpipe_containment_counts(want, 2, want_table)
    assert np.array_equal(hits, whits) and np.array_equal(sizes, wsizes)
    assert set(np.argsort(-hits[-1].astype(np.int64))[:4]) == {1, 5, 7, 30}
    kc.free()
    table.free()


def test_three_hundred_samples_through_one_set_of_counters(hip, oracle_lib):
    """The entry counters are not zeroed between samples: their words carry a pass number of eight bits (mg_kcount_core.h:
    kc_entry_count) and are zeroed when it wraps.  300 resets of one set of counters, two samples in turn — one that saturates its
    k-mers, one that barely touches them: every sample's counts are its own, before, at and after the wrap."""
    ks = [31, 51]
    rng = np.random.default_rng(6500)
    gb, go = random_genomes(rng, 8, 8000)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], 300)
    table = hip.refdb_build(h, khi, klo, o, ks)
    want_table = oracle_lib.refpipe_build(h, khi, klo, o, ks)
    table.index_kmers()
    samples = []
    for present, n in (([1, 4], 6000), ([6], 300)):
        rb, ro, _ = sample_reads(rng, gb, go, n, 150, err=0.01, present=present)
        want, _ = oracle_lib.refpipe_count_kmers(rb, ro, ks[-1], want_table["kmer_hi"], want_table["kmer_lo"], cs=3)
        samples.append((hip.array(np.concatenate([rb, np.zeros(64, np.uint8)])), hip.array(ro), len(ro) - 1, int(ro[-1]), want))
    assert samples[0][4].max() == 3 and 0 < samples[1][4].sum() < samples[0][4].sum()
    kc = table.kmer_counts()
    for i in range(300):
        d_b, d_o, n, nb, want = samples[i & 1]
        kc.reset()
        kc.add_dev(d_b.ptr, d_o.ptr, n, nb)
        if i < 4 or i % 37 == 0 or 250 <= i <= 262 or i >= 296:
            assert np.array_equal(kc.download(), want), i
    kc.free()
    table.free()
