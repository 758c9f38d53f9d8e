"""GPU parity of the REFERENCE PIPELINE (stage A/B as scripts/select_db.py:50-59,73-76 wires KMC and CMash: k_max-mers only on
the read side, smaller-k columns from prefixes of the matched k_max-mers) against the oracle, through the C ABI.
Bit-exact: integer work throughout."""
import numpy as np
import pytest

from util import flat, refpipe_case

pytestmark = pytest.mark.gpu

K_SETS = [[21, 31, 51], [30, 40, 50, 60], [4, 6, 10], [5, 33, 64], [32], [8, 16, 32], [31, 32, 33, 64]]


@pytest.fixture(params=[0, 1], ids=["canonical_kmer_hash", "cmash_recollection"])
def mode(request, hip, oracle_lib):
    hip.set_hash_mode(request.param)
    oracle_lib.set_hash_mode(request.param)
    yield request.param
    hip.set_hash_mode(0)
    oracle_lib.set_hash_mode(0)


def _same_table(got, want, ks, kmers=True):
    assert np.array_equal(got["pair_hash"], want["pair_hash"])
    assert np.array_equal(got["pair_gen"], want["pair_gen"])
    assert np.array_equal(got["gsize"], want["gsize"])
    if kmers:
        assert np.array_equal(got["kmer_hi"], want["kmer_hi"]) and np.array_equal(got["kmer_lo"], want["kmer_lo"])
    for k in ks[:-1]:
        g, w = got["small"][k], want["small"][k]
        assert g["nprefix"] == w["nprefix"], k
        for key in ("pa", "pb", "cid", "cgen", "gsize"):
            assert np.array_equal(g[key], w[key]), (k, key)


@pytest.mark.parametrize("ks", K_SETS, ids=str)
def test_table_and_query_match_the_oracle(hip, oracle_lib, mode, ks):
    rng = np.random.default_rng(9000 + 13 * sum(ks) + mode)
    genomes, reads = refpipe_case(rng)
    _table_and_query(hip, oracle_lib, ks, genomes, reads, "canonical")


@pytest.mark.parametrize("ks", [[21, 31, 51], [30, 40, 50, 60], [4, 6, 10], [32]], ids=str)
def test_forward_selected_table_and_query_match_the_oracle(hip, oracle_lib, mode, ks):
    """`build_db --sketch_hash forward` (mg_sketch_genomes_kmers_forward): entries chosen by the hash of the k-mer as it stands and kept
    as they stand; one genome holds a stretch and its reverse complement — the same matching identity twice in its sketch."""
    rng = np.random.default_rng(9400 + 13 * sum(ks) + mode)
    genomes, reads = refpipe_case(rng)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    genomes[3] = genomes[3].upper()[:500] + genomes[3].upper()[100:400].translate(comp)[::-1]
    for a in range(0, len(genomes[3]) - 150, 40):
        reads += [genomes[3][a:a + 150]] * 2
    h, o = _table_and_query(hip, oracle_lib, ks, genomes, reads, "forward")
    if ks[-1] >= 21:
        ids = [int(x) for x in h[int(o[3]):int(o[4])]]
        assert len(ids) > len(set(ids))


def _table_and_query(hip, oracle_lib, ks, genomes, reads, sketch_hash):
    kmax, n = ks[-1], 150
    gb, go = flat(genomes)
    # stage A' with the k-mers kept
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, kmax, n, sketch_hash=sketch_hash)
    oh, ohi, olo, oo = oracle_lib.sketch_genomes_kmers(gb, go, kmax, n, sketch_hash=sketch_hash)
    assert np.array_equal(o, oo) and np.array_equal(h, oh)
    assert np.array_equal(khi, ohi) and np.array_equal(klo, olo)
    # the table, built on the device
    table = hip.refdb_build(h, khi, klo, o, ks)
    want = oracle_lib.refpipe_build(oh, ohi, olo, oo, ks)
    got = table.download()
    _same_table(got, want, ks)
    assert table.max_hash == (int(h.max()) if len(h) else 0) and table.ngenomes == len(genomes)
    # the query: stage A at k_max only, with and without the table's pre-filter
    rb, ro = flat(reads)
    d_b, d_o = hip.array(rb), hip.array(ro)
    filt = hip.filter_build(h)
    for use_filter in (False, True):
        sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(reads), kmax, table.max_hash, 0, filt if use_filter else None)
        qh, qc = sk.download()
        for ci in (1, 2, 3):
            hits, sizes = hip.refpipe_containment(sk, table, ci)
            whits, wsizes = oracle_lib.refpipe_containment(qh, qc, ci, want)
            assert np.array_equal(hits, whits), (ks, ci, use_filter)
            assert np.array_equal(sizes, wsizes)
        # the largest k's column is the plain stage B against the same table
        phits, psizes = hip.containment(sk, table.kmax_table(), 2)
        hits2, sizes2 = hip.refpipe_containment(sk, table, 2)
        assert np.array_equal(hits2[-1], phits) and np.array_equal(sizes2[-1], psizes)
        sk.free()
    # the oracle's own unfiltered sketch gives the same columns (every table hash passes its own filter)
    uh, uc, _, _ = oracle_lib.sketch_reads(rb, ro, kmax, hmax=table.max_hash)
    whits, _ = oracle_lib.refpipe_containment(uh, uc, 2, want)
    assert np.array_equal(hits2, whits)
    assert all(whits[ki][g] > 0 for ki in range(len(ks)) for g in (0, 2, 4))
    # a table uploaded from the stored arrays answers the same
    up = hip.refdb_upload(ks, len(genomes), got["pair_hash"], got["pair_gen"], got["gsize"], table.max_hash, [got["small"][k] for k in ks[:-1]])
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(reads), kmax, up.max_hash, 0, filt)
    hits3, sizes3 = hip.refpipe_containment(sk, up, 2)
    assert np.array_equal(hits3, hits2) and np.array_equal(sizes3, sizes2)
    _same_table(up.download(kmers=False), want, ks, kmers=False)
    for x in (sk, up, table, filt, d_b, d_o):
        x.free()
    return h, o


@pytest.mark.parametrize("world", [2, 3, 8])
def test_rank_shares_of_the_table_add_up(hip, oracle_lib, mode, world):
    """A multi-GPU job's layout on one GPU: rank r holds the pairs of a hash range (with their pa / pb) and the run of every
    count list that falls in a prefix range; mark on every share, OR the bitmaps, count on every share, sum: the unsharded columns."""
    ks = [21, 31, 51]
    rng = np.random.default_rng(77 + world + mode)
    genomes, reads = refpipe_case(rng, ngenomes=7)
    gb, go = flat(genomes)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], 120)
    table = hip.refdb_build(h, khi, klo, o, ks)
    full = table.download()
    G, npairs = len(genomes), len(full["pair_hash"])
    rb, ro = flat(reads)
    d_b, d_o = hip.array(rb), hip.array(ro)
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(reads), ks[-1], table.max_hash, 0)
    want_hits, want_sizes = hip.refpipe_containment(sk, table, 2)
    cuts = [npairs * r // world for r in range(world + 1)]
    shares = []
    for r in range(world):
        a, b = cuts[r], cuts[r + 1]
        small = []
        for k in ks[:-1]:
            t = full["small"][k]
            lo, hi = t["nprefix"] * r // world, t["nprefix"] * (r + 1) // world
            ca, cb = np.searchsorted(t["cid"], lo, side="left"), np.searchsorted(t["cid"], hi, side="left")
            small.append(dict(pa=t["pa"][a:b], pb=t["pb"][a:b], cid=t["cid"][ca:cb], cgen=t["cgen"][ca:cb],
                              gsize=np.bincount(t["cgen"][ca:cb], minlength=G).astype(np.uint32), nprefix=t["nprefix"]))
        shares.append(hip.refdb_upload(ks, G, full["pair_hash"][a:b], full["pair_gen"][a:b],
                                       np.bincount(full["pair_gen"][a:b], minlength=G).astype(np.uint32), table.max_hash, small))
    # mark on every share; the OR of the bitmaps
    d_hs = hip.empty(2 * G, np.uint32)
    hits = np.zeros((len(ks), G), np.uint64)
    sizes = np.zeros((len(ks), G), np.uint64)
    ored = None
    for sh in shares:
        hip.refpipe_mark_dev(sk, sh, 2, d_hs.ptr, d_hs.ptr + 4 * G)
        a = d_hs.download()
        hits[-1] += a[:G]
        sizes[-1] += a[G:2 * G]
        words = []
        for ki in range(len(ks) - 1):
            ptr, nw = sh.marks(ki)
            buf = np.zeros(max(nw, 1), np.uint32)
            hip._chk(hip.lib.mg_memcpy_d2h(buf.ctypes.data_as(__import__("ctypes").c_void_p), __import__("ctypes").c_void_p(ptr), __import__("ctypes").c_uint64(4 * nw)))
            words.append(buf[:nw])
        ored = words if ored is None else [x | y for x, y in zip(ored, words)]
    d_marks = [hip.array(w if len(w) else np.zeros(1, np.uint32)) for w in ored]
    d_out = hip.empty(2 * G * (len(ks) - 1), np.uint32)
    for sh in shares:
        hip.refpipe_count_dev(sh, [m.ptr for m in d_marks], [d_out.ptr + 4 * (2 * ki * G) for ki in range(len(ks) - 1)],
                              [d_out.ptr + 4 * ((2 * ki + 1) * G) for ki in range(len(ks) - 1)])
        a = d_out.download().reshape(len(ks) - 1, 2, G)
        hits[:-1] += a[:, 0, :]
        sizes[:-1] += a[:, 1, :]
    assert np.array_equal(hits, want_hits) and np.array_equal(sizes, want_sizes)
    for x in shares + d_marks + [sk, table, d_b, d_o, d_hs, d_out]:
        x.free()


def test_edges(hip, oracle_lib, mode):
    """No genomes; genomes without a single k-mer; no reads; a sketch that matches nothing; a bottom-s sketch is refused."""
    from metalign_amd import _hip
    ks = [21, 31, 51]
    empty = hip.refdb_build(np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(1, np.uint64), ks)
    assert empty.ngenomes == 0 and empty.sizes() == (0, [0, 0], [0, 0])
    gb, go = flat([b"ACGT" * 3, b"", b"NNNN"])
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, 51, 100)
    assert len(h) == 0 and list(o) == [0, 0, 0, 0]
    t0 = hip.refdb_build(h, khi, klo, o, ks)
    rng = np.random.default_rng(3)
    genomes, reads = refpipe_case(rng, strains=False)
    gb, go = flat(genomes)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, 51, 100)
    table = hip.refdb_build(h, khi, klo, o, ks)
    want = oracle_lib.refpipe_build(*oracle_lib.sketch_genomes_kmers(gb, go, 51, 100), ks)
    for rd in ([], [b"ACGTN" * 40] * 2):
        rb, ro = flat(rd)
        d_b, d_o = hip.array(rb if len(rb) else np.zeros(1, np.uint8)), hip.array(ro)
        sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(rd), 51, table.max_hash, 0)
        hits, sizes = hip.refpipe_containment(sk, table, 2)
        assert not hits.any() and np.array_equal(sizes, oracle_lib.refpipe_containment(np.zeros(0, np.uint64), np.zeros(0, np.uint32), 2, want)[1])
        hz, sz = hip.refpipe_containment(sk, t0, 2)
        assert hz.shape == (3, 3) and not hz.any() and not sz.any()
        sk.free(); d_b.free(); d_o.free()
    rb, ro = flat(reads)
    d_b, d_o = hip.array(rb), hip.array(ro)
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(reads), 51, table.max_hash, 20)  # bottom-20: truncated
    assert sk.truncated
    with pytest.raises(_hip.HipError):
        hip.refpipe_containment(sk, table, 2)
    for x in (sk, d_b, d_o, table, t0, empty):
        x.free()


def test_a_table_that_goes_up_beside_the_reads(hip, oracle_lib):
    """mg_refdb_upload_begin (refdb_upload(..., wait=False): what select_main does): the handle comes back while the library's
    threads still copy the arrays up; max_hash / ngenomes answer at once; the first call that reads the table waits for it and
    checks it — the same columns as the table uploaded at once, and a corrupt table is refused THERE."""
    rng = np.random.default_rng(4242)
    genomes, reads = refpipe_case(rng, ngenomes=40, glen=(20_000, 30_000))
    ks, n = [21, 31, 51], 2000
    gb, go = flat(genomes)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], n)
    built = hip.refdb_build(h, khi, klo, o, ks)
    got = built.download(kmers=False)
    rb, ro = flat(reads)
    d_b, d_o = hip.array(rb), hip.array(ro)
    small = [got["small"][k] for k in ks[:-1]]
    for rep in range(3):  # (several: the uploader's slots and threads are made and joined every time)
        up = hip.refdb_upload(ks, len(genomes), got["pair_hash"], got["pair_gen"], got["gsize"], built.max_hash, small, wait=False)
        assert up.max_hash == built.max_hash and up.ngenomes == len(genomes)
        sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(reads), ks[-1], up.max_hash, 0, None)  # (the stream's place: work beside the upload)
        hits, sizes = hip.refpipe_containment(sk, up, 2)
        whits, wsizes = hip.refpipe_containment(sk, built, 2)
        assert np.array_equal(hits, whits) and np.array_equal(sizes, wsizes)
        same = up.download(kmers=False)
        assert np.array_equal(same["pair_hash"], got["pair_hash"]) and np.array_equal(same["small"][ks[0]]["cid"], got["small"][ks[0]]["cid"])
        sk.free()
        up.free()
    # freed before anything read it: the uploader is joined, nothing is left behind
    hip.refdb_upload(ks, len(genomes), got["pair_hash"], got["pair_gen"], got["gsize"], built.max_hash, small, wait=False).free()
    # a corrupt table: accepted by begin, refused by the first reader (and by mg_refdb_upload at once)
    bad_gen = got["pair_gen"].copy()
    bad_gen[len(bad_gen) // 2] = len(genomes) + 7
    bad = hip.refdb_upload(ks, len(genomes), got["pair_hash"], bad_gen, got["gsize"], built.max_hash, small, wait=False)
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(reads), ks[-1], built.max_hash, 0, None)
    with pytest.raises(Exception, match="corrupt"):
        hip.refpipe_containment(sk, bad, 2)
    bad.free()
    with pytest.raises(Exception, match="corrupt"):
        hip.refdb_upload(ks, len(genomes), got["pair_hash"], bad_gen, got["gsize"], built.max_hash, small)
    bad_pa = dict(small[0])
    bad_pa["pa"] = small[0]["pa"].copy()
    bad_pa["pa"][3] = 0xFFFFFFF0
    bad = hip.refdb_upload(ks, len(genomes), got["pair_hash"], got["pair_gen"], got["gsize"], built.max_hash, [bad_pa] + small[1:], wait=False)
    with pytest.raises(Exception, match="corrupt"):
        bad.kmax_table()
    bad.free()
    for x in (sk, built, d_b, d_o):
        x.free()
