"""-m gpu: the HIP path (through the C ABI) against the CPU oracle and the reference's golden vectors."""
import numpy as np
import pytest

import stage_c_checks as sc
import util
from metalign_amd import _hip

pytestmark = pytest.mark.gpu

U64_MAX = 0xFFFFFFFFFFFFFFFF


@pytest.mark.parametrize("k", [1, 4, 11, 16, 21, 31, 32, 33, 48, 51, 60, 64])
def test_sketch_reads_every_kmer(hip, oracle_lib, k):
    """hmax = max: the sketch is the full distinct k-mer hash set with counts; bit-exact."""
    rng = np.random.default_rng(100 + k)
    gb, go = util.random_genomes(rng, 3, 4000, with_n=True)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 700, 150, err=0.02, ragged=(k % 2 == 0), lower=True)
    h, c, trunc, seen = hip.sketch_reads(bases, offsets, k)
    oh, oc, otrunc, oseen = oracle_lib.sketch_reads(bases, offsets, k)
    assert seen == oseen
    assert trunc == otrunc is False
    assert np.array_equal(h, oh) and np.array_equal(c, oc)
    assert np.all(h[1:] > h[:-1])


@pytest.mark.parametrize("k,frac,s", [(21, 0.02, 0), (21, 0.3, 500), (31, 0.05, 0), (51, 0.1, 64), (60, 1.0, 1000)])
def test_sketch_reads_threshold_and_truncation(hip, oracle_lib, k, frac, s):
    rng = np.random.default_rng(7 * k)
    gb, go = util.random_genomes(rng, 5, 6000)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 3000, 150, err=0.01)
    hmax = U64_MAX if frac >= 1.0 else int(frac * 2**64)
    h, c, trunc, seen = hip.sketch_reads(bases, offsets, k, hmax=hmax, s=s)
    oh, oc, otrunc, oseen = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmax, s=s)
    assert (trunc, seen) == (otrunc, oseen)
    assert np.array_equal(h, oh) and np.array_equal(c, oc)
    assert c.max() >= 2  # coverage is high enough that the count>=2 path is exercised


def test_sketch_reads_edge_inputs(hip, oracle_lib):
    k = 21
    # empty read set
    h, c, trunc, seen = hip.sketch_reads(np.zeros(0, np.uint8), np.zeros(1, np.uint64), k)
    assert h.size == 0 and seen == 0
    # reads shorter than k, empty reads, a read of exactly k, all-N read
    seqs = [b"", b"ACGT", b"ACGTACGTACGTACGTACGTA", b"N" * 40, b"ACGTACGTACGTACGTACGTAN" + b"ACGTTGCATGCATGCATGCATGCAA", b""]
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8)
    offsets = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
    h, c, trunc, seen = hip.sketch_reads(bases, offsets, k)
    oh, oc, _, oseen = oracle_lib.sketch_reads(bases, offsets, k)
    assert seen == oseen and np.array_equal(h, oh) and np.array_equal(c, oc)
    # long reads (tile larger than the LDS stage -> direct path) mixed with short ones
    rng = np.random.default_rng(5)
    gb, go = util.random_genomes(rng, 2, 30000)
    b1, o1, _ = util.sample_reads(rng, gb, go, 70, 3000, err=0.0)
    h, c, _, seen = hip.sketch_reads(b1, o1, 31, hmax=int(0.2 * 2**64))
    oh, oc, _, oseen = oracle_lib.sketch_reads(b1, o1, 31, hmax=int(0.2 * 2**64))
    assert seen == oseen and np.array_equal(h, oh) and np.array_equal(c, oc)


def test_sketch_reverse_complement_invariance(hip):
    """Size-independent property: a read set and its reverse complement have the same sketch."""
    rng = np.random.default_rng(11)
    gb, go = util.random_genomes(rng, 4, 20000)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 20000, 150, err=0.0)
    rc = bases.copy()
    for i in range(len(offsets) - 1):
        a, b = int(offsets[i]), int(offsets[i + 1])
        rc[a:b] = util._COMP[bases[a:b][::-1]]
    for k in (21, 51):
        h1, c1, _, s1 = hip.sketch_reads(bases, offsets, k, hmax=2**60)
        h2, c2, _, s2 = hip.sketch_reads(rc, offsets, k, hmax=2**60)
        assert s1 == s2 and np.array_equal(h1, h2) and np.array_equal(c1, c2)


@pytest.mark.parametrize("k,n", [(21, 1000), (31, 200), (51, 1000), (60, 50), (5, 1000)])
def test_sketch_genomes_matches_oracle(hip, oracle_lib, k, n):
    rng = np.random.default_rng(k + n)
    lens = [12000, 0, 37, 5000, k - 1 if k > 1 else 0, 9000]
    bases = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), p=[.2495, .2495, .2495, .2495, .002], size=sum(lens)).astype(np.uint8)
    offsets = np.cumsum([0] + lens).astype(np.uint64)
    h, o = hip.sketch_genomes(bases, offsets, k, n)
    oh, oo = oracle_lib.sketch_genomes(bases, offsets, k, n)
    assert np.array_equal(o, oo) and np.array_equal(h, oh)


@pytest.mark.parametrize("k,s,ci", [(21, 0, 2), (21, 3000, 1), (31, 0, 2), (51, 800, 3)])
def test_containment_matches_oracle(hip, oracle_lib, k, s, ci):
    rng = np.random.default_rng(31 + k)
    gb, go = util.random_genomes(rng, 40, 8000)
    dbh, dbo = oracle_lib.sketch_genomes(gb, go, k, 300)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 6000, 150, err=0.01, present=[1, 5, 9, 33])
    hmax = int(dbh.max())
    table = hip.upload_table(dbh, dbo)
    assert table.max_hash == hmax
    d_b, d_o = hip.array(bases), hip.array(offsets)
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(offsets) - 1, k, hmax, s)
    hits, sizes = hip.containment(sk, table, ci)
    qh, qc = sk.download()
    oh, oc, otrunc, _ = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmax, s=s)
    assert np.array_equal(qh, oh) and np.array_equal(qc, oc) and sk.truncated == otrunc
    ohits, osizes = oracle_lib.containment(oh, oc, otrunc, ci, dbh, dbo)
    assert np.array_equal(hits, ohits) and np.array_equal(sizes, osizes)
    if s == 0:
        ci_vals = hits / np.maximum(sizes, 1)
        assert ci_vals[[1, 5, 9, 33]].min() > 0.5 and np.delete(ci_vals, [1, 5, 9, 33]).max() < 0.1


@pytest.mark.parametrize("cs", [0, 1, 3, 7])
def test_count_saturation_matches_oracle(hip, oracle_lib, cs):
    """Counters saturate at cs (kmc -cs3, select_db.py:50): count = min(occurrences, cs), on the counting-table path
    (look before add, clamp at pack), on the list path and through a merge; cs = 0 keeps exact counts."""
    rng = np.random.default_rng(11)
    gb, go = util.random_genomes(rng, 3, 6000)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 9000, 150, err=0.005)  # ~75x coverage: most counts far above cs
    k = 21
    d_b, d_o = hip.array(bases), hip.array(offsets)
    hip.count_saturation(cs)
    try:
        assert hip.count_saturation() == cs
        for hmax in (U64_MAX, int(0.01 * 2 ** 64)):  # the table path, and few enough candidates for the list path
            sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(offsets) - 1, k, hmax, 0)
            qh, qc = sk.download()
            oh, oc, _, _ = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmax, cs=cs)
            assert np.array_equal(qh, oh) and np.array_equal(qc, oc)
            assert int(qc.max()) == (cs if cs else int(oc.max())) and (cs == 0 or int(oc.max()) == cs)
            # merged with itself: min(2 c, cs)
            both_h, both_c = np.concatenate([qh, qh]), np.concatenate([qc, qc])
            d_h, d_c = hip.array(both_h), hip.array(both_c)
            for lo, hi in ((0, int(qh[-1])), (1, 0)):  # through the counting table / through the sorting merge
                m = hip.sketch_merge_dev(d_h.ptr, d_c.ptr, both_h.size, k, lo, hi)
                mh, mc = m.download()
                want = 2 * qc.astype(np.uint64)
                assert np.array_equal(mh, qh) and np.array_equal(mc, np.minimum(want, cs) if cs else want)
        if cs:
            table = hip.upload_table(np.sort(rng.choice(qh, size=50, replace=False)), np.array([0, 50], dtype=np.uint64))
            with pytest.raises(_hip.HipError):
                hip.containment(sk, table, cs + 1)  # a threshold above the saturation could never be met
    finally:
        hip.count_saturation(3)


@pytest.mark.parametrize("ks", [(21, 31, 51), (30, 40, 50, 60), (21, 31), (15, 21, 31, 51)])
@pytest.mark.parametrize("kind", ["clean", "ragged", "dirty"])
def test_multi_k_sketch_matches_oracle_per_k(hip, oracle_lib, ks, kind, monkeypatch, knobs):
    """mg_sketch_reads_multi_dev_async: every k of the query from one pass (select_db.py:73-76 is a multi-k query).
    {21,31,51} and {30,40,50,60} run the fused kernel (one roller at the largest k, every smaller k's k-mer derived
    from it), other sets one launch per k; either way each sketch equals the oracle's for that k bit for bit — hashes,
    saturated counts, k-mers seen — with and without the table's membership filter, on the three tile walks (equal
    reads / ragged incl. shorter than every k and empty / N and lower case), and equals the single-k entry point."""
    rng = np.random.default_rng(sum(ks) + len(kind))
    gb, go = util.random_genomes(rng, 12, 9000)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 9000, 150, err=0.01, ragged=(kind != "clean"), lower=(kind == "dirty"))
    if kind == "dirty":
        bases = bases.copy()
        bases[rng.integers(0, bases.size, size=bases.size // 300)] = ord("N")
    if kind != "clean":  # reads shorter than some / every k, and empty ones
        lens = np.diff(offsets.astype(np.int64))
        lens[rng.integers(0, len(lens), size=200)] = rng.integers(0, 61, size=200)
        o2 = np.zeros(len(lens) + 1, dtype=np.uint64)
        o2[1:] = np.cumsum(lens)
        keep = np.concatenate([np.arange(int(offsets[i]), int(offsets[i]) + int(lens[i])) for i in range(len(lens))])
        bases, offsets = bases[keep], o2
    nreads = len(offsets) - 1
    d_b, d_o = hip.array(bases), hip.array(offsets)
    tables = [oracle_lib.sketch_genomes(gb, go, k, 400)[0] for k in ks]
    for filtered in (False, True):
        hmaxs = [int(0.2 * 2 ** 64)] * len(ks) if not filtered else [int(t.max()) for t in tables]
        filts = [hip.filter_build(t) for t in tables] if filtered else None
        sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, nreads, list(ks), hmaxs, 0, filts)
        for i, k in enumerate(ks):
            h, c = sks[i].download()
            if filtered:
                oh, oc, _, oseen = oracle_lib.sketch_reads_filtered(bases, offsets, k, tables[i], hmax=hmaxs[i])
            else:
                oh, oc, _, oseen = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmaxs[i])
            assert np.array_equal(h, oh) and np.array_equal(c, oc), (k, filtered)
            assert sks[i].kmers_seen == oseen, (k, sks[i].kmers_seen, oseen)
            one = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, k, hmaxs[i], 0, filt=filts[i] if filtered else None)
            h1, c1 = one.download()
            assert np.array_equal(h, h1) and np.array_equal(c, c1)
    # an undersized counting table of the fused launch is detected per k and that sketch is redone (list path)
    knobs("distinct_hint_ppm", 500)
    sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, nreads, list(ks), [int(0.2 * 2 ** 64)] * len(ks), 0, None)
    rebuilt = [sk.resolve() for sk in sks]
    knobs("distinct_hint_ppm", 0)
    assert any(rebuilt)
    for i, k in enumerate(ks):
        h, c = sks[i].download()
        oh, oc, _, _ = oracle_lib.sketch_reads(bases, offsets, k, hmax=int(0.2 * 2 ** 64))
        assert np.array_equal(h, oh) and np.array_equal(c, oc), k


@pytest.mark.gpu
@pytest.mark.parametrize("order", ["filter", "slot"])
def test_flush_order_does_not_change_the_sketch(hip, oracle_lib, order, monkeypatch, knobs):
    """A flush of stage A's candidate buffer looks at the membership filter and at the candidate's home slot in the
    counting table; each wavefront takes whichever order was cheaper for its previous flush (mg_sketch_multi.hip:
    MultiSink::flush).  Pinned to either order (knob flush_order, a test hook) the sketches are the oracle's, bit for
    bit, at high coverage (nearly every candidate is a repeat) and at none (every candidate is new), with error k-mers
    the filter rejects, for the single-k kernel and the fused one."""
    knobs("flush_order", {"f": 1, "s": 2}.get(order[:1], 0))
    rng = np.random.default_rng(5 + len(order))
    gb, go = util.random_genomes(rng, 10, 8000)
    ks = (21, 31, 51)
    tables = [oracle_lib.sketch_genomes(gb, go, k, 800)[0] for k in ks]
    filts = [hip.filter_build(t) for t in tables]
    hmaxs = [int(t.max()) for t in tables]
    for nreads, present in ((12000, 2), (600, 10)):  # ~110x coverage of two genomes; 1x of all ten
        pick = np.sort(rng.choice(10, size=present, replace=False))
        bases, offsets, _ = util.sample_reads(rng, gb, go, nreads, 150, err=0.02, present=pick)
        d_b, d_o = hip.array(bases), hip.array(offsets)
        sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, nreads, list(ks), hmaxs, 0, filts)
        for i, k in enumerate(ks):
            h, c = sks[i].download()
            oh, oc, _, oseen = oracle_lib.sketch_reads_filtered(bases, offsets, k, tables[i], hmax=hmaxs[i])
            assert np.array_equal(h, oh) and np.array_equal(c, oc) and sks[i].kmers_seen == oseen, (k, nreads)
            one = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, k, hmaxs[i], 0, filt=filts[i])
            h1, c1 = one.download()
            assert np.array_equal(h1, oh) and np.array_equal(c1, oc), (k, nreads)
        # and without a filter (the order then only decides when the slot is read)
        one = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, 21, hmaxs[0], 0)
        h1, c1 = one.download()
        oh, oc, _, _ = oracle_lib.sketch_reads(bases, offsets, 21, hmax=hmaxs[0])
        assert np.array_equal(h1, oh) and np.array_equal(c1, oc)


@pytest.mark.parametrize("ks", [(21, 31, 51), (30, 40, 50, 60)])
def test_multi_k_sketch_long_reads_take_the_hbm_walk(hip, oracle_lib, ks):
    """Tiles that do not fit the LDS stage (contigs instead of reads) are walked straight out of HBM by the fused kernel
    too; mixed with ordinary reads, N runs and lower case."""
    rng = np.random.default_rng(len(ks))
    lens = np.concatenate([rng.integers(30_000, 90_000, size=5), rng.integers(0, 200, size=300), [70_000]])
    rng.shuffle(lens)
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    bases = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(offsets[-1])).astype(np.uint8)
    bases[rng.integers(0, bases.size, size=bases.size // 400)] = ord("N")
    low = rng.random(bases.size) < 0.05
    bases[low] |= 0x20
    d_b, d_o = hip.array(bases), hip.array(offsets)
    hmaxs = [int(0.25 * 2 ** 64)] * len(ks)
    sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, len(lens), list(ks), hmaxs, 0, None)
    for i, k in enumerate(ks):
        h, c = sks[i].download()
        oh, oc, _, oseen = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmaxs[i])
        assert np.array_equal(h, oh) and np.array_equal(c, oc) and sks[i].kmers_seen == oseen, k


def test_sketch_merge_equals_single_pass(hip, oracle_lib):
    """Two read shards sketched separately and merged == one pass over all reads (the multi-GPU merge)."""
    rng = np.random.default_rng(77)
    gb, go = util.random_genomes(rng, 6, 9000)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 4000, 150, err=0.01)
    k, hmax = 21, int(0.25 * 2**64)
    for s in (0, 700):
        half = 2000
        cut = int(offsets[half])
        parts = [(bases[:cut], offsets[: half + 1]), (bases[cut:], offsets[half:] - offsets[half])]
        hs, cs, truncs = [], [], []
        for b, o in parts:
            h, c, t, _ = hip.sketch_reads(b, o, k, hmax=hmax, s=s)
            hs.append(h); cs.append(c); truncs.append(t)
        allh, allc = np.concatenate(hs), np.concatenate(cs)
        bound = min([h[-1] for h, t in zip(hs, truncs) if t] or [U64_MAX])
        d_h, d_c = hip.array(allh), hip.array(allc)
        merged = hip.sketch_from_pairs_dev(d_h.ptr, d_c.ptr, allh.size, k, s=s, any_truncated=any(truncs), bound=int(bound))
        mh, mc = merged.download()
        oh, oc, otrunc, _ = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmax, s=s)
        assert np.array_equal(mh, oh) and np.array_equal(mc, oc)
        assert merged.truncated == otrunc


def test_deferred_merge_equals_merge(hip):
    """mg_sketch_merge_dev_async (nothing synchronised, handle pending) == mg_sketch_merge_dev == numpy: the table
    path, its consumption by stage B while pending, and the redo when a pair lies outside the declared hash range."""
    rng = np.random.default_rng(5)
    lo, hi = 1 << 40, (1 << 58) - 1
    n = 300_000
    pool = rng.integers(lo, hi, size=120_000, dtype=np.uint64)
    h = pool[rng.integers(0, len(pool), size=n)]
    c = rng.integers(1, 5, size=n).astype(np.uint32)
    want_h, inv = np.unique(h, return_inverse=True)
    want_exact = np.bincount(inv, weights=c).astype(np.uint32)
    want_c = np.minimum(want_exact, hip.count_saturation())  # counters saturate at cs = 3 (kmc -cs3), sums included
    d_h, d_c = hip.array(h), hip.array(c)
    # a small table to run stage B against while the merged sketch is still pending
    G, per = 40, 500
    dbh = np.sort(rng.choice(want_h, size=(G, per)), axis=1).reshape(-1)
    dbo = np.arange(G + 1, dtype=np.uint64) * np.uint64(per)
    table = hip.upload_table(dbh, dbo)
    d_hits, d_sizes = hip.empty(G, np.uint32), hip.empty(G, np.uint32)
    for declared, redo in (((lo, hi), False), ((lo, (lo + hi) // 2), True)):
        sk = hip.sketch_merge_dev_async(d_h.ptr, d_c.ptr, n, 21, declared[0], declared[1])
        hip.containment_dev(sk, table, 2, d_hits.ptr, d_sizes.ptr)  # consumed on the device, nothing settled yet
        rebuilt = sk.resolve()
        assert bool(rebuilt) == redo
        if rebuilt:  # what was derived from the pending handle is stale by contract: again
            hip.containment_dev(sk, table, 2, d_hits.ptr, d_sizes.ptr)
        hip.sync()
        gh, gc = sk.download()
        assert np.array_equal(gh, want_h) and np.array_equal(gc, want_c)
        good = set(want_h[want_c >= 2].tolist())
        hits = d_hits.download()
        for g in (0, 7, 39):
            assert hits[g] == len({x for x in dbh[g * per:(g + 1) * per].tolist() if x in good} ) or \
                hits[g] == sum(1 for x in dbh[g * per:(g + 1) * per].tolist() if x in good)
        ref = hip.sketch_merge_dev(d_h.ptr, d_c.ptr, n, 21, declared[0], declared[1])
        rh, rc = ref.download()
        assert np.array_equal(rh, gh) and np.array_equal(rc, gc)
        ref.free()
        sk.free()
    for b in (d_h, d_c, d_hits, d_sizes):
        b.free()
    table.free()


@pytest.mark.parametrize("name,idx,run", sc.hand_cases(), ids=lambda v: str(v) if not isinstance(v, dict) else "")
def test_stage_c_hand_cases_hip(hip, name, idx, run, monkeypatch, knobs, tmp_path):
    sc.check_hand_case(name, run, None, monkeypatch, tmp_path)


@pytest.mark.parametrize("name", ["single_3k", "paired_2k", "single_100k", "paired_40k"])
def test_stage_c_bulk_hip(hip, name, monkeypatch, knobs, tmp_path):
    spec = sc.load_bulk()[name]
    sam, dbp = sc.materialise_bulk(name, spec, tmp_path)
    for run in spec["runs"]:
        sc.check_run(sam, dbp, run, None, monkeypatch, tmp_path, mm_digest_only=True)


@pytest.mark.parametrize("force_hashed", [False, True])
def test_stage_c_records_vs_oracle_large_taxa(hip, oracle_lib, force_hashed, monkeypatch, knobs):
    """Random record streams straight into the C ABI: direct LDS bins at three, then two workgroups per CU (37 / 3500
    taxa) and, beyond 4096 taxa, hashed bins (open addressing, 2048 slots) that overflow into the workgroups' private bins;
    and the hashed bins on the small taxonomies too (knob k3_hashed, a test hook): few taxa — every probe hits at
    once — and 3500 — the table is crowded, the probe limit drops and many taxa overflow."""
    if force_hashed:
        knobs("k3_hashed", 1)
    rng = np.random.default_rng(9)
    for ntax, nref in ((37, 90), (3500, 7000), (5000, 9000)):
        n = 300000
        ref2tax = rng.integers(0, ntax, size=nref).astype(np.uint32)
        recs = np.zeros(n, dtype=oracle_lib.REC_DTYPE)
        new = rng.random(n) < 0.7
        new[0] = True
        hot = rng.integers(0, nref, size=20)
        ref = np.where(rng.random(n) < 0.6, hot[rng.integers(0, 20, size=n)], rng.integers(0, nref, size=n))
        recs["ref_new"] = ref.astype(np.uint32) | (new.astype(np.uint32) << 31)
        total = rng.integers(30, 151, size=n).astype(np.uint32)
        recs["total"] = total
        recs["matched"] = (total * np.clip(rng.normal(0.8, 0.25, size=n), 0, 1)).astype(np.uint32)
        flags = rng.choice([0, 16, 256, 272, 2048, 99, 147, 355, 403, 65, 129, 73, 137], size=n,
                           p=[.3, .3, .1, .1, .02, .03, .03, .02, .02, .02, .02, .02, .02]).astype(np.uint32)
        seqlen = np.where(rng.random(n) < 0.8, total, 0).astype(np.uint32)
        recs["flag_len"] = flags | (seqlen << 12)
        got = hip.profile_assign(recs, ref2tax, ntax, 0.5)
        want = oracle_lib.profile_assign(recs, ref2tax, ntax, 0.5)
        for key in want:
            assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), key


@pytest.mark.parametrize("force_hashed", [False, True])
@pytest.mark.parametrize("flush_tiles,grid", [(1, 0), (2, 3), (3, 7), (0, 2)])
def test_stage_c_bins_flushed_in_the_middle_of_a_workgroups_life(hip, oracle_lib, force_hashed, flush_tiles, grid, monkeypatch, knobs):
    """A workgroup's LDS bins are flushed every 256 (direct) / 15 (hashed) tiles — which no input of a test's size reaches, a
    workgroup living for a tile or two.  With the test hooks (fewer workgroups, shorter intervals) the same records go through
    mid-life flushes (hashed: into the per-XCD copies, keys released and claimed again), bins that persist over many tiles,
    and the last dump of a table that was just flushed; small, crowded (3500 taxa in 2048 slots) and large taxonomies."""
    if force_hashed:
        knobs("k3_hashed", 1)
    if flush_tiles:
        knobs("k3_flush_tiles", flush_tiles)
    if grid:
        knobs("k3_grid", grid)
    rng = np.random.default_rng(21)
    for ntax, nref in ((37, 90), (3500, 7000), (5000, 9000)):
        n = 60000  # 30 tiles
        ref2tax = rng.integers(0, ntax, size=nref).astype(np.uint32)
        recs = np.zeros(n, dtype=oracle_lib.REC_DTYPE)
        new = rng.random(n) < 0.75
        new[0] = True
        hot = rng.integers(0, nref, size=12)
        ref = np.where(rng.random(n) < 0.5, hot[rng.integers(0, 12, size=n)], rng.integers(0, nref, size=n))
        recs["ref_new"] = ref.astype(np.uint32) | (new.astype(np.uint32) << 31)
        total = rng.integers(30, 151, size=n).astype(np.uint32)
        recs["total"] = total
        recs["matched"] = (total * np.clip(rng.normal(0.8, 0.25, size=n), 0, 1)).astype(np.uint32)
        recs["flag_len"] = rng.choice([0, 16, 256, 272, 99, 147], size=n, p=[.35, .35, .1, .1, .05, .05]).astype(np.uint32) | (total << 12)
        want = oracle_lib.profile_assign(recs, ref2tax, ntax, 0.5)
        for rep in range(2):  # (twice: the copies a pass leaves behind must be as it found them)
            got = hip.profile_assign(recs, ref2tax, ntax, 0.5)
            for key in want:
                assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), (ntax, rep, key)


def test_sketch_two_million_distinct(hip, oracle_lib):
    """~2.4M distinct hashes at hmax = max (bucket path, thousands of buckets) against the oracle's full sort."""
    rng = np.random.default_rng(99)
    gb, go = util.random_genomes(rng, 40, 60000)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 20000, 150, err=0.0)
    oh, oc, _, oseen = oracle_lib.sketch_reads(bases, offsets, 31)
    h, c, _, seen = hip.sketch_reads(bases, offsets, 31)
    assert seen == oseen and np.array_equal(h, oh) and np.array_equal(c, oc)


def test_sketch_bucket_path_overflow_falls_back(hip, oracle_lib, monkeypatch, knobs):
    """A k-mer repeated far more often than a bucket slab holds (adapter-like reads) forces the list path;
    the knob force_list exercises the list path on ordinary input.  Same sketch either way."""
    rng = np.random.default_rng(123)
    gb, go = util.random_genomes(rng, 8, 30000)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 30000, 150, err=0.005)
    k = 21
    oh, oc, _, oseen = oracle_lib.sketch_reads(bases, offsets, k, hmax=int(0.3 * 2**64))
    for force in (False, True):
        if force:
            knobs("force_list", 1)
        h, c, _, seen = hip.sketch_reads(bases, offsets, k, hmax=int(0.3 * 2**64))
        assert seen == oseen and np.array_equal(h, oh) and np.array_equal(c, oc)
    knobs("force_list", 0)
    # an undersized table (distinct-count hint far too low) must be detected and redone on the list path
    knobs("distinct_hint_ppm", 500)
    h, c, _, seen = hip.sketch_reads(bases, offsets, k, hmax=int(0.3 * 2**64))
    assert seen == oseen and np.array_equal(h, oh) and np.array_equal(c, oc)
    knobs("distinct_hint_ppm", 0)
    # 6000 copies of one read: each of its k-mers occurs 6000x (one table slot, count 6000+)
    rep = np.tile(bases[: 150], 6000)
    b2 = np.concatenate([bases, rep])
    o2 = np.concatenate([offsets, offsets[-1] + (np.arange(1, 6001, dtype=np.uint64) * np.uint64(150))])
    oh, oc, _, oseen = oracle_lib.sketch_reads(b2, o2, k, hmax=int(0.3 * 2**64))
    h, c, _, seen = hip.sketch_reads(b2, o2, k, hmax=int(0.3 * 2**64))
    assert seen == oseen and np.array_equal(h, oh) and np.array_equal(c, oc)
    assert c.max() == 3  # saturated (kmc -cs3); with exact counts the repeated read's k-mers count 6000+
    hip.count_saturation(0)
    try:
        oh, oc, _, _ = oracle_lib.sketch_reads(b2, o2, k, hmax=int(0.3 * 2**64), cs=0)
        h, c, _, _ = hip.sketch_reads(b2, o2, k, hmax=int(0.3 * 2**64))
        assert np.array_equal(h, oh) and np.array_equal(c, oc) and c.max() >= 6000
    finally:
        hip.count_saturation(3)


def test_deferred_sketch_feeds_containment_without_a_sync(hip, oracle_lib, monkeypatch, knobs):
    """mg_sketch_reads_dev_async: stage B consumes the sketch while its size is still on the device; the results
    equal the oracle's, and a counting-table overflow discovered at resolve() is reported so that stage B is redone."""
    rng = np.random.default_rng(77)
    gb, go = util.random_genomes(rng, 12, 30000)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 120000, 150, err=0.02)  # many distinct (erroneous) k-mers
    k, n = 21, 300
    dbh, dbo = oracle_lib.sketch_genomes(gb, go, k, n)
    hmax = int(dbh.max())
    oh, oc, otr, oseen = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmax)
    ohits, osizes = oracle_lib.containment(oh, oc, otr, 2, dbh, dbo)
    table = hip.upload_table(dbh, dbo)
    d_b, d_o = hip.array(bases), hip.array(offsets)
    d_h, d_s = hip.empty(12, np.uint32), hip.empty(12, np.uint32)
    for hint, want_rebuilt in ((None, False), ("0.0005", True)):
        if hint:
            knobs("distinct_hint_ppm", int(float(hint) * 1e6))
        sk = hip.sketch_reads_dev_async(d_b.ptr, d_o.ptr, len(offsets) - 1, k, hmax, 0)
        hip.containment_dev(sk, table, 2, d_h.ptr, d_s.ptr)  # queued behind stage A, nothing synchronised yet
        hip.sync()
        rebuilt = sk.resolve()
        assert rebuilt or not want_rebuilt  # (the adaptive table sizing may also overflow on its own: same handling)
        if rebuilt:
            hip.containment_dev(sk, table, 2, d_h.ptr, d_s.ptr)
        assert np.array_equal(d_h.download(), ohits) and np.array_equal(d_s.download(), osizes)
        h, c = sk.download()
        assert sk.kmers_seen == oseen and np.array_equal(h, oh) and np.array_equal(c, oc)
        assert not sk.resolve()  # idempotent
        sk.free()
        knobs("distinct_hint_ppm", 0)


@pytest.mark.parametrize("name", ["single_3k", "paired_2k", "single_100k", "paired_40k"])
def test_device_multimap_matches_reference_cami(hip, name, tmp_path):
    """--device_multimap: resolve_multi_prop runs on the GPU over the resident CSR.  The CAMI
    profile equals the reference's golden text up to the 1e-6 abundance tolerance (sums are order-dependent), and
    the exceptions the reference raises still surface."""
    from metalign_amd import map_and_profile as mp
    spec = sc.load_bulk()[name]
    sam, dbp = sc.materialise_bulk(name, spec, tmp_path)
    for run in spec["runs"]:
        out = str(tmp_path / "abund_dev.tsv")
        args = sc.make_args(sam, dbp, out, dict(run["args"], device_multimap=True))
        raised = None
        try:
            mp.map_main(args)
        except SystemExit as e:
            raised = ["SystemExit", str(e)]
        except Exception as e:  # noqa: BLE001
            raised = [type(e).__name__, str(e)]
        assert raised == run.get("map_main_raises")
        if raised is None:
            with open(out) as fh:
                sc._cami_close(fh.read(), run["cami"])


def test_device_multimap_equals_host_resolution(hip, oracle_lib):
    """mg_profile_resolve_multimapped_dev vs the vectorised host version on random multimapped lists with repeated
    and dropped taxa, with and without length normalisation (1e-12 relative)."""
    from metalign_amd import map_and_profile as mp
    import argparse
    rng = np.random.default_rng(5)
    nrec, nref, ntax = 300000, 400, 37
    recs = np.zeros(nrec, dtype=oracle_lib.REC_DTYPE)
    new = rng.random(nrec) < 0.45
    new[0] = True
    recs["ref_new"] = rng.integers(0, nref, size=nrec).astype(np.uint32) | (new.astype(np.uint32) << 31)
    recs["total"] = 100
    recs["matched"] = rng.integers(30, 101, size=nrec)
    recs["flag_len"] = rng.choice([0, 16, 256, 272], size=nrec).astype(np.uint32) | (np.uint32(100) << 12)
    ref2tax = rng.integers(0, ntax, size=nref).astype(np.uint32)
    taxids = ["t%d" % i for i in range(ntax)]
    res_host = hip.profile_assign(recs, ref2tax, ntax, 0.5)
    res_dev = hip.profile_assign_resident(recs, ref2tax, ntax, 0.5)
    assert res_dev["mm_nreads"] == len(res_host["mm_hitlen"]) > 1000
    for key in ("count", "bases", "first_seen"):
        assert np.array_equal(res_dev[key], res_host[key])
    taxid2info = {t: [1000.0 + 13 * i] for i, t in enumerate(taxids)}
    for norm in (False, True):
        args = argparse.Namespace(verbose=False, length_normalize=norm, low_mem=False)
        keep = [i for i in range(ntax) if res_host["count"][i] > 0 and i % 5 != 0]  # some taxa dropped by a cutoff
        a = {taxids[i]: [int(res_host["count"][i]), float(res_host["bases"][i])] for i in keep}
        b = {k: list(v) for k, v in a.items()}
        mp.resolve_multi_prop_csr(args, a, dict(res_host, taxids=taxids), taxid2info)
        mp.resolve_multi_prop_device(args, b, dict(res_dev, taxids=taxids), taxid2info)
        assert a.keys() == b.keys()
        changed = 0
        for k in a:
            assert abs(a[k][1] - b[k][1]) <= 1e-12 * max(1.0, abs(a[k][1])), (k, a[k], b[k])
            changed += a[k][1] != float(res_host["bases"][taxids.index(k)])
        assert changed > 5
    res_dev["resident"].free()


@pytest.mark.parametrize("k,s", [(21, 0), (31, 0), (21, 500)])
def test_filtered_sketch_matches_oracle_and_keeps_containment(hip, oracle_lib, k, s):
    """mg_sketch_reads_filtered_dev: the sketch restricted to hashes whose bit is set in the table's membership
    filter (the reference's `-f ...bf` role).  Bit-exact against the oracle's restatement, sync and deferred; and
    containment is what the unfiltered sketch gives, because every table hash passes its own filter."""
    rng = np.random.default_rng(31 + k + s)
    gb, go = util.random_genomes(rng, 10, 4000)          # tiny genomes: the threshold alone filters little
    bases, offsets, _ = util.sample_reads(rng, gb, go, 20000, 150, err=0.01)
    n = 800
    dbh, dbo = oracle_lib.sketch_genomes(gb, go, k, n)
    hmax = int(dbh.max())
    filt = hip.filter_build(dbh)
    assert filt.log2_bits == 17  # 8000 hashes x 16 = 128000 -> 2^17
    d_b, d_o = hip.array(bases), hip.array(offsets)
    table = hip.upload_table(dbh, dbo)
    oh, oc, otr, oseen = oracle_lib.sketch_reads_filtered(bases, offsets, k, dbh, hmax=hmax, s=s)
    uh, uc, utr, _ = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmax, s=0)
    assert len(oh) < 0.5 * len(uh)                        # the filter does remove most of the sketch here
    for deferred in (False, True):
        fn = hip.sketch_reads_dev_async if deferred else hip.sketch_reads_dev
        sk = fn(d_b.ptr, d_o.ptr, len(offsets) - 1, k, hmax, s, filt=filt)
        h, c = sk.download()
        assert np.array_equal(h, oh) and np.array_equal(c, oc)
        assert sk.truncated == otr and sk.kmers_seen == oseen
        if s == 0:
            hits, sizes = hip.containment(sk, table, 2)
            ohits, osizes = oracle_lib.containment(uh, uc, utr, 2, dbh, dbo)   # from the UNFILTERED oracle sketch
            assert np.array_equal(hits, ohits) and np.array_equal(sizes, osizes)
        sk.free()
    # the list path (small / forced) applies the same filter
    from metalign_amd import _hip
    _hip.debug_set("force_list", 1)
    try:
        sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(offsets) - 1, k, hmax, s, filt=filt)
        h, c = sk.download()
        assert np.array_equal(h, oh) and np.array_equal(c, oc)
        sk.free()
    finally:
        _hip.debug_set("force_list", 0)
    # an empty table: nothing passes
    empty = hip.filter_build(np.zeros(0, dtype=np.uint64))
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(offsets) - 1, k, hmax, 0, filt=empty)
    assert sk.size == 0
    sk.free()


@pytest.fixture
def cmash_mode(hip, oracle_lib):
    """Hash definition 1 in the library AND the oracle (min(hash(kmer), hash(revcomp)) % 9999999999971: CMash as SURVEY.md
    §8(c) recollects it, unverified); back to the default afterwards."""
    hip.set_hash_mode(1)
    oracle_lib.set_hash_mode(1)
    yield
    hip.set_hash_mode(0)
    oracle_lib.set_hash_mode(0)


def test_cmash_recollection_mode_matches_the_oracle(hip, oracle_lib, cmash_mode):
    """Stage A', A (one k and the fused launches, table and list paths, the three tile walks) and B under hash mode 1 ==
    the oracle under the same mode; and the two modes give different sketches of the same reads."""
    rng = np.random.default_rng(4242)
    gb, go = util.random_genomes(rng, 12, 4000, with_n=True)
    for ks in ([21], [21, 31, 51], [30, 40, 50, 60], [1], [32, 33], [64]):
        tabs = []
        for k in ks:
            dbh, dbo = hip.sketch_genomes(gb, go, k, 150)
            odbh, odbo = oracle_lib.sketch_genomes(gb, go, k, 150)
            assert np.array_equal(dbh, odbh) and np.array_equal(dbo, odbo), k
            assert int(dbh.max()) < oracle_lib.CMASH_PRIME
            tabs.append((dbh, dbo))
        for ragged, with_n in ((False, False), (True, False), (False, True)):
            rb, ro, _ = util.sample_reads(rng, gb, go, 9000, 150, ragged=ragged, lower=True, present=[2, 7])
            if with_n:
                rb[rng.integers(0, rb.size, size=50)] = ord("N")
            d_b, d_o = hip.array(rb), hip.array(ro)
            hmaxs = [int(t[0].max()) for t in tabs]
            sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, len(ro) - 1, ks, hmaxs, 0, None)
            for k, sk, (dbh, dbo), hm in zip(ks, sks, tabs, hmaxs):
                sk.resolve()
                qh, qc = sk.download()
                oh, oc, otr, _ = oracle_lib.sketch_reads(rb, ro, k, hmax=hm)
                assert np.array_equal(qh, oh) and np.array_equal(qc, oc), (ks, k, ragged, with_n)
                table = hip.upload_table(dbh, dbo)
                hits, sizes = hip.containment(sk, table, 2)
                ohits, osizes = oracle_lib.containment(oh, oc, otr, 2, dbh, dbo)
                assert np.array_equal(hits, ohits) and np.array_equal(sizes, osizes)
                table.free()
                sk.free()
            d_b.free()
            d_o.free()
    # the list path (tiny input) and a different sketch than mode 0's
    rb, ro, _ = util.sample_reads(rng, gb, go, 40, 150, present=[2])
    h1, c1, _, _ = hip.sketch_reads(rb, ro, 21)
    oh, oc, _, _ = oracle_lib.sketch_reads(rb, ro, 21)
    assert np.array_equal(h1, oh) and np.array_equal(c1, oc)
    hip.set_hash_mode(0)
    h0, _, _, _ = hip.sketch_reads(rb, ro, 21)
    hip.set_hash_mode(1)
    assert len(h0) == len(h1) and not np.array_equal(h0, h1)


def test_cmash_prefix_tables_match_the_oracle(hip, oracle_lib, cmash_mode):
    """The k < k_max tables of hash mode 1 — the k-prefixes of the sketched k_max-mers (mg_sketch_genomes_prefix) — against
    the oracle, and a query against them: every read k-mer is a candidate there (the table's largest key is near the prime),
    the membership filter does the rejecting, and the containment counts equal the oracle's."""
    rng = np.random.default_rng(777)
    gb, go = util.random_genomes(rng, 10, 3000, with_n=True)
    n = 120
    for kmax, ks in ((60, [30, 40, 50]), (51, [21, 31]), (33, [32]), (64, [1, 64])):
        for k in ks:
            h, o = hip.sketch_genomes_prefix(gb, go, kmax, k, n)
            oh, oo = oracle_lib.sketch_genomes_prefix(gb, go, kmax, k, n)
            assert np.array_equal(h, oh) and np.array_equal(o, oo), (kmax, k)
    # the stock query shape: K = {30,40,50,60}, prefix tables for 30 / 40 / 50, the sketch itself at 60
    ks, kmax = [30, 40, 50, 60], 60
    tabs = [hip.sketch_genomes_prefix(gb, go, kmax, k, n) if k < kmax else hip.sketch_genomes(gb, go, k, n) for k in ks]
    want_last = oracle_lib.sketch_genomes(gb, go, kmax, n)
    assert np.array_equal(tabs[-1][0], want_last[0])
    rb, ro, _ = util.sample_reads(rng, gb, go, 6000, 150, lower=True, present=[1, 6])
    d_b, d_o = hip.array(rb), hip.array(ro)
    hmaxs = [int(t[0].max()) for t in tabs]
    assert hmaxs[0] > oracle_lib.CMASH_PRIME // 2 > hmaxs[-1]  # prefix keys are not bottom-n values: no threshold to speak of
    filts = [hip.filter_build(t[0]) for t in tabs]
    sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, len(ro) - 1, ks, hmaxs, 0, filts)
    for k, sk, (dbh, dbo), hm in zip(ks, sks, tabs, hmaxs):
        sk.resolve()
        qh, qc = sk.download()
        oh, oc, otr, _ = oracle_lib.sketch_reads_filtered(rb, ro, k, dbh, hmax=hm)
        assert np.array_equal(qh, oh) and np.array_equal(qc, oc), k
        table = hip.upload_table(dbh, dbo)
        hits, sizes = hip.containment(sk, table, 2)
        ohits, osizes = oracle_lib.containment(oh, oc, otr, 2, dbh, dbo)
        assert np.array_equal(hits, ohits) and np.array_equal(sizes, osizes), k
        assert hits[1] > 0.5 * sizes[1] and hits[6] > 0.5 * sizes[6] and hits[0] < 0.2 * max(sizes[0], 1)
        table.free()
        sk.free()
    d_b.free()
    d_o.free()


def _exact_sketch(oracle_lib, bases, offsets, k, table_hashes, hmax, cs=3):
    """What a sketch made against a table's RESIDENT INDEX holds: the read k-mers that are hashes of the table."""
    uh, uc, _, seen = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmax, s=0)
    keep = np.isin(uh, table_hashes)
    return uh[keep], uc[keep], seen


@pytest.mark.parametrize("ks", [(21, 31, 51), (30, 40, 50, 60), (25, 33)])
def test_resident_index_sketches_are_the_exact_intersection(hip, oracle_lib, ks):
    """mg_filter_make_resident: the counting table seeded once with every hash of the genome table, counters tagged with
    the epoch of the call.  A sketch made with it is the oracle's unfiltered sketch restricted to the table's hashes
    (counts saturating at 3 as ever) — for the fused launch, the one-k kernel and a k set without a fused kernel, call
    after call on DIFFERENT reads (a counter of an earlier epoch reads as zero), and containment is what the
    unfiltered sketch gives."""
    rng = np.random.default_rng(sum(ks))
    gb, go = util.random_genomes(rng, 12, 6000)
    tabs = [oracle_lib.sketch_genomes(gb, go, k, 900) for k in ks]
    hmaxs = [int(t[0].max()) for t in tabs]
    filts = [hip.filter_build(t[0]) for t in tabs]
    spread = 1 if len(ks) == 2 else 0  # (spread 1: buckets planned for twice the hashes — half the load)
    for f, t, hm in zip(filts, tabs, hmaxs):
        assert f.resident_bytes == 0
        assert f.make_resident(t[0], hm, spread)
        assert f.resident_bytes > 0
    tables = [hip.upload_table(*t) for t in tabs]
    for rep, (nreads, present, err) in enumerate(((9000, 3, 0.02), (700, 12, 0.0), (9000, 5, 0.03), (64, 1, 0.0))):
        pick = np.sort(rng.choice(12, size=present, replace=False))
        bases, offsets, _ = util.sample_reads(rng, gb, go, nreads, 150, err=err, present=pick)
        d_b, d_o = hip.array(bases), hip.array(offsets)
        sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, nreads, list(ks), hmaxs, 0, filts)
        for i, k in enumerate(ks):
            eh, ec, seen = _exact_sketch(oracle_lib, bases, offsets, k, tabs[i][0], hmaxs[i])
            h, c = sks[i].download()
            assert np.array_equal(h, eh) and np.array_equal(c, ec) and sks[i].kmers_seen == seen, (k, rep)
            one = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, k, hmaxs[i], 0, filt=filts[i])
            h1, c1 = one.download()
            assert np.array_equal(h1, eh) and np.array_equal(c1, ec), (k, rep)
            # a threshold above the table's largest hash changes nothing; a bottom-s sketch keeps the bit filter's definition
            cut = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, k, (1 << 64) - 1, 0, filt=filts[i])
            h2, c2 = cut.download()
            assert np.array_equal(h2, eh) and np.array_equal(c2, ec)
            cut.free()
            cut = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, k, hmaxs[i], 50, filt=filts[i])
            h2, c2 = cut.download()
            fh, fc, ftr, _ = oracle_lib.sketch_reads_filtered(bases, offsets, k, tabs[i][0], hmax=hmaxs[i], s=50)
            assert np.array_equal(h2, fh) and np.array_equal(c2, fc) and cut.truncated == ftr
            hits, sizes = hip.containment(sks[i], tables[i], 2)
            uh, uc, utr, _ = oracle_lib.sketch_reads(bases, offsets, k, hmax=hmaxs[i], s=0)
            ohits, osizes = oracle_lib.containment(uh, uc, utr, 2, *tabs[i])
            assert np.array_equal(hits, ohits) and np.array_equal(sizes, osizes), (k, rep)
            for x in (one, cut):
                x.free()
        for x in sks:
            x.free()
    # exact counts (saturation off): the epoch's first candidate starts the counter at one
    hip.count_saturation(0)
    try:
        bases, offsets, _ = util.sample_reads(rng, gb, go, 20000, 150, err=0.0, present=np.array([4]))
        d_b, d_o = hip.array(bases), hip.array(offsets)
        for _ in range(2):
            sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, 20000, ks[0], hmaxs[0], 0, filt=filts[0])
            h, c = sk.download()
            uh, uc, _, _ = oracle_lib.sketch_reads(bases, offsets, ks[0], hmax=hmaxs[0], s=0, cs=0)
            keep = np.isin(uh, tabs[0][0])
            assert np.array_equal(h, uh[keep]) and np.array_equal(c, uc[keep]) and c.max() > 50
            sk.free()
    finally:
        hip.count_saturation(3)
    # hashes that crowd one range: no index, the filter stays what it was
    crowded = hip.filter_build(tabs[0][0])
    assert not crowded.make_resident(np.arange(5000, dtype=np.uint64), (1 << 63))
    assert crowded.resident_bytes == 0
    for f in filts + [crowded]:
        f.free()


@pytest.mark.parametrize("hook", ["distinct_HINT", "resident_scan"])
def test_resident_index_list_overflow_and_slot_walk(hip, oracle_lib, hook, monkeypatch, knobs):
    """The sketch of a resident index is made from the LIST of hashes the kernel touched.  A list (or sketch buffer) sized for
    far fewer hashes than the sample has (knob distinct_hint_ppm, a test hook) is reported like a table overflow and the
    sketch made again at full size when it is resolved — exact either way; and a walk over every slot of the index
    (knob resident_scan) gives the same sketch as the list."""
    knobs("distinct_hint_ppm" if hook.endswith("HINT") else hook, 100 if hook.endswith("HINT") else 1)
    rng = np.random.default_rng(77)
    ks = (21, 31, 51)
    gb, go = util.random_genomes(rng, 40, 6000)
    tabs = [oracle_lib.sketch_genomes(gb, go, k, 900) for k in ks]
    hmaxs = [int(t[0].max()) for t in tabs]
    filts = [hip.filter_build(t[0]) for t in tabs]
    for f, t, hm in zip(filts, tabs, hmaxs):
        assert f.make_resident(t[0], hm)
    bases, offsets, _ = util.sample_reads(rng, gb, go, 30000, 150, err=0.01)
    d_b, d_o = hip.array(bases), hip.array(offsets)
    for _ in range(2):
        sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, 30000, list(ks), hmaxs, 0, filts)
        for i, k in enumerate(ks):
            eh, ec, seen = _exact_sketch(oracle_lib, bases, offsets, k, tabs[i][0], hmaxs[i])
            rebuilt = sks[i].resolve()
            if hook.endswith("HINT"):
                assert rebuilt
            h, c = sks[i].download()
            assert len(eh) > 20000 and np.array_equal(h, eh) and np.array_equal(c, ec), k
            one = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, 30000, k, hmaxs[i], 0, filt=filts[i])
            h1, c1 = one.download()
            assert np.array_equal(h1, eh) and np.array_equal(c1, ec), k
            one.free()
        for x in sks:
            x.free()
    for f in filts:
        f.free()


def test_resident_index_with_passes_in_flight_on_two_streams(hip, oracle_lib):
    """Two sketch calls in flight on the library's two stage-A streams count in two COPIES of the index (the second made on
    that stream's first use), each call under an epoch of its own; a third call reuses the first stream's copy while the
    other stream's sketch is still unresolved.  Every sketch is the exact intersection of ITS reads with the table."""
    rng = np.random.default_rng(2025)
    ks = (21, 31, 51)
    gb, go = util.random_genomes(rng, 30, 6000)
    tabs = [oracle_lib.sketch_genomes(gb, go, k, 900) for k in ks]
    hmaxs = [int(t[0].max()) for t in tabs]
    filts = [hip.filter_build(t[0]) for t in tabs]
    for f, t, hm in zip(filts, tabs, hmaxs):
        assert f.make_resident(t[0], hm)
    one_copy = [f.resident_bytes for f in filts]
    samples = []
    for present in (np.arange(0, 10), np.arange(10, 30), np.arange(5, 12)):
        bases, offsets, _ = util.sample_reads(rng, gb, go, 40000, 150, err=0.01, present=present)
        samples.append((bases, offsets, hip.array(bases), hip.array(offsets)))
    try:
        queued = []
        for which, (bases, offsets, d_b, d_o) in zip((1, 2, 1), samples):
            hip.stage_a_side_stream(which)
            queued.append(hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, 40000, list(ks), hmaxs, 0, filts))
    finally:
        hip.stage_a_side_stream(0)
    assert [f.resident_bytes for f in filts] == [2 * b for b in one_copy]
    for (bases, offsets, _, _), sks in zip(samples, queued):
        for i, k in enumerate(ks):
            eh, ec, seen = _exact_sketch(oracle_lib, bases, offsets, k, tabs[i][0], hmaxs[i])
            h, c = sks[i].download()
            assert np.array_equal(h, eh) and np.array_equal(c, ec) and sks[i].kmers_seen == seen, k
            sks[i].free()
    for f in filts:
        f.free()


def test_resident_index_under_the_cmash_hash_definition(hip, oracle_lib, cmash_mode):
    """Hash definition 1 (hashes below 9999999999971) with PREFIX tables for the smaller k — every read k-mer is a candidate
    there, the regime the resident index is for: the fused launch, the one-k kernel and a k set without a fused kernel
    against the exact intersection of the oracle's sketch (same definition) with the table."""
    rng = np.random.default_rng(1971)
    gb, go = util.random_genomes(rng, 10, 5000)
    for ks in ((21, 31, 51), (25, 35)):  # (hash mode 1 is built for a list of k: mg_sketch_cmash.hip)
        full = oracle_lib.sketch_genomes(gb, go, ks[-1], 400)
        tabs = [oracle_lib.sketch_genomes_prefix(gb, go, ks[-1], k, 400) for k in ks[:-1]] + [full]
        hmaxs = [int(t[0].max()) for t in tabs]
        assert hmaxs[0] > 0.5 * oracle_lib.CMASH_PRIME  # (a prefix table's largest key is near the prime)
        filts = [hip.filter_build(t[0]) for t in tabs]
        for f, t, hm in zip(filts, tabs, hmaxs):
            assert f.make_resident(t[0], hm, 1)
        for nreads, present in ((8000, [1, 4]), (500, list(range(10)))):
            bases, offsets, _ = util.sample_reads(rng, gb, go, nreads, 150, err=0.01, present=present)
            d_b, d_o = hip.array(bases), hip.array(offsets)
            sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, nreads, list(ks), hmaxs, 0, filts)
            for i, k in enumerate(ks):
                eh, ec, seen = _exact_sketch(oracle_lib, bases, offsets, k, tabs[i][0], hmaxs[i])
                h, c = sks[i].download()
                assert len(eh) > 0 and np.array_equal(h, eh) and np.array_equal(c, ec) and sks[i].kmers_seen == seen, (ks, k)
                one = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, k, hmaxs[i], 0, filt=filts[i])
                h1, c1 = one.download()
                assert np.array_equal(h1, eh) and np.array_equal(c1, ec), (ks, k)
                one.free()
                sks[i].free()
        for f in filts:
            f.free()


def test_dropping_a_resident_index_leaves_the_bit_filter(hip, oracle_lib):
    """mg_filter_drop_resident (a job drops the index when its priming passes measure the bit filter faster — a sample most of
    whose candidates are NOT hashes of the table): the memory is given back and sketches are the filter's again."""
    rng = np.random.default_rng(31337)
    gb, go = util.random_genomes(rng, 10, 6000)
    k = 31
    dbh, dbo = oracle_lib.sketch_genomes(gb, go, k, 900)
    hmax = int(dbh.max())
    bases, offsets, _ = util.sample_reads(rng, gb, go, 20000, 150, err=0.01)
    d_b, d_o = hip.array(bases), hip.array(offsets)
    filt = hip.filter_build(dbh)
    assert filt.make_resident(dbh, hmax)
    eh, ec, _ = _exact_sketch(oracle_lib, bases, offsets, k, dbh, hmax)
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, 20000, k, hmax, 0, filt=filt)
    h, c = sk.download()
    assert np.array_equal(h, eh) and np.array_equal(c, ec)
    sk.free()
    filt.drop_resident()
    assert filt.resident_bytes == 0
    fh, fc, _, _ = oracle_lib.sketch_reads_filtered(bases, offsets, k, dbh, hmax=hmax)
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, 20000, k, hmax, 0, filt=filt)
    h, c = sk.download()
    assert len(fh) > len(eh) and np.array_equal(h, fh) and np.array_equal(c, fc)
    sk.free()
    filt.drop_resident()  # (nothing to drop: fine)
    filt.free()
