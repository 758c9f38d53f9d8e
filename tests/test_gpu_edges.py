"""-m gpu: degenerate inputs through the C ABI (empty, single element, extreme k) against the oracle."""
import numpy as np
import pytest

import util
from metalign_amd import _hip

pytestmark = pytest.mark.gpu
U64_MAX = 0xFFFFFFFFFFFFFFFF


def test_empty_and_tiny_record_streams(hip, oracle_lib):
    r2t = np.array([0, 1, 1], dtype=np.uint32)
    for n in (0, 1, 2, 3):
        recs = np.zeros(n, dtype=oracle_lib.REC_DTYPE)
        if n:
            recs["ref_new"] = np.array([1 | 0x80000000, 2, 1 | 0x80000000][:n], dtype=np.uint32)
            recs["total"] = 10
            recs["matched"] = 10
            recs["flag_len"] = (10 << 12)
        got = hip.profile_assign(recs, r2t, 2, 0.5)
        want = oracle_lib.profile_assign(recs, r2t, 2, 0.5)
        for key in want:
            assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), (n, key)


def test_one_giant_read_group(hip, oracle_lib):
    """A single read with 5000 alignment lines (one thread walks it; paired-end intersection is quadratic)."""
    rng = np.random.default_rng(1)
    n = 5000
    r2t = rng.integers(0, 7, size=30).astype(np.uint32)
    for flags in ([0, 256], [99, 147, 355, 403]):
        recs = np.zeros(n + 2, dtype=oracle_lib.REC_DTYPE)
        recs["ref_new"] = rng.integers(0, 30, size=n + 2).astype(np.uint32)
        recs["ref_new"][[0, 1, n + 1]] |= 0x80000000  # starter read (1 line), the giant read, a closing read
        recs["total"] = 100
        recs["matched"] = rng.integers(30, 101, size=n + 2)
        recs["flag_len"] = rng.choice(flags, size=n + 2).astype(np.uint32) | (100 << 12)
        got = hip.profile_assign(recs, r2t, 7, 0.5)
        want = oracle_lib.profile_assign(recs, r2t, 7, 0.5)
        for key in want:
            assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), key


@pytest.mark.parametrize("hashed", [False, True])
def test_stage_c_reads_around_the_bin_length_limit(hip, oracle_lib, monkeypatch, knobs, hashed):
    """Uniquely mapped reads of 2^14 bases or more bypass the workgroup's bins (one queue word holds 14 bits of length):
    lengths either side of the limit and up to the record field's 2^20 - 1, through the direct and the hashed bins."""
    if hashed:
        knobs("k3_hashed", 1)
    rng = np.random.default_rng(7)
    n = 60000
    ntax = 37
    r2t = rng.integers(0, ntax, size=200).astype(np.uint32)
    recs = np.zeros(n, dtype=oracle_lib.REC_DTYPE)
    new = rng.random(n) < 0.8
    new[0] = True
    recs["ref_new"] = rng.integers(0, 200, size=n).astype(np.uint32) | (new.astype(np.uint32) << 31)
    lens = rng.choice([150, 16383, 16384, 16385, 500000, (1 << 20) - 1], size=n, p=[0.5, 0.1, 0.1, 0.1, 0.1, 0.1]).astype(np.uint32)
    recs["total"] = lens
    recs["matched"] = lens - rng.integers(0, 20, size=n).astype(np.uint32)
    recs["flag_len"] = rng.choice([0, 16, 256], size=n, p=[0.5, 0.4, 0.1]).astype(np.uint32) | (lens << 12)
    got = hip.profile_assign(recs, r2t, ntax, 0.5)
    want = oracle_lib.profile_assign(recs, r2t, ntax, 0.5)
    assert int(np.asarray(want["count"]).sum()) > n // 3
    for key in want:
        assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), key


def test_sketch_degenerate_inputs(hip, oracle_lib):
    cases = {
        "all N": ([b"N" * 200] * 70, 21),
        "k=1": ([b"ACGTTGCAAC", b"", b"T"], 1),
        "k=64 exact": ([b"ACGT" * 16, b"ACGT" * 16 + b"A"], 64),
        "one read": ([b"ACGTACGTTGCATGCATGCAAGCTAGCT"], 21),
        "homopolymer x 5000": ([b"A" * 150] * 5000, 31),
    }
    for name, (seqs, k) in cases.items():
        bases = np.frombuffer(b"".join(seqs), dtype=np.uint8) if any(seqs) else np.zeros(0, np.uint8)
        offs = np.cumsum([0] + [len(s) for s in seqs]).astype(np.uint64)
        h, c, t, seen = hip.sketch_reads(bases, offs, k)
        oh, oc, ot, oseen = oracle_lib.sketch_reads(bases, offs, k)
        assert (seen, t) == (oseen, ot), name
        assert np.array_equal(h, oh) and np.array_equal(c, oc), name


def test_containment_with_empty_operands(hip, oracle_lib):
    dbh = np.array([5, 9, 100, 7], dtype=np.uint64)
    dbo = np.array([0, 3, 3, 4], dtype=np.uint64)  # genome 1 has an empty sketch
    table = hip.upload_table(dbh, dbo)
    assert table.ngenomes == 3 and table.max_hash == 100
    empty = hip.sketch_reads_dev(hip.array(np.zeros(1, np.uint8)).ptr, hip.array(np.zeros(1, np.uint64)).ptr, 0, 21, U64_MAX, 0)
    hits, sizes = hip.containment(empty, table, 2)
    assert list(hits) == [0, 0, 0] and list(sizes) == [3, 0, 1]
    qh, qc = np.array([7, 9, 50], dtype=np.uint64), np.array([2, 1, 9], dtype=np.uint32)
    d_h, d_c = hip.array(qh), hip.array(qc)
    sk = hip.sketch_from_pairs_dev(d_h.ptr, d_c.ptr, 3, 21)
    hits, sizes = hip.containment(sk, table, 2)
    oh, os_ = oracle_lib.containment(qh, qc, False, 2, dbh, dbo)
    assert np.array_equal(hits, oh) and np.array_equal(sizes, os_)


def test_bad_arguments_fail_loudly(hip):
    with pytest.raises(_hip.HipError):
        hip.sketch_reads(np.frombuffer(b"ACGT", np.uint8), np.array([0, 4], np.uint64), 65)
    with pytest.raises(_hip.HipError):
        hip.sketch_reads(np.frombuffer(b"ACGT", np.uint8), np.array([0, 4], np.uint64), 0)
    with pytest.raises(_hip.HipError):
        hip.sketch_genomes(np.frombuffer(b"ACGT", np.uint8), np.array([0, 4], np.uint64), 21, 0)


def test_pipelined_passes_hold_steady_memory(hip):
    """Hundreds of pipelined passes (stage A and stage C on the library's side streams: freed blocks are fenced before
    reuse, mg_core.hip) must not keep acquiring device memory: free memory after 150 passes and after 750 differs by
    less than a handful of blocks.  (Sealing a batch of fenced frees only when a free list ran empty let the held memory
    grow with the square root of the pass count — found by tools/soak.py.)"""
    from metalign_amd import synth
    from metalign_amd.distributed import ShardJob
    gb, go = synth.make_genomes(60, 20000)
    rb, ro, src = synth.make_reads(gb, go, 100000, npresent=9)
    recs = synth.make_alignment_records(src + 1, 61)
    dbh, dbo = hip.sketch_genomes(gb, go, 21, 300)
    job = ShardJob(hip, None, 0, 1, k=21)
    job.load(rb, ro, recs, np.arange(61, dtype=np.uint32), dbh, dbo)
    first = job.run(150, want_multimapped=True)
    hip.sync()
    free0, total, pooled0 = hip.mem_info()
    last = job.run(600, want_multimapped=True)
    hip.sync()
    free1, _, pooled1 = hip.mem_info()
    assert total > 0 and np.array_equal(first["hits"], last["hits"]) and np.array_equal(first["count"], last["count"])
    assert free0 - free1 < (8 << 20), (free0, free1, pooled0, pooled1)


def test_pipelined_run_equals_single_steps(hip, oracle_lib):
    """ShardJob.run(n) on a single shard: pass i+1 is queued before pass i is read back (two result sets, one marker
    per pass); every pass gives what a stand-alone step gives (and what the oracle gives).  The exchange-path
    pipelining (stage A on its own stream) is covered by tests/dist_single_rank.py."""
    from metalign_amd import synth
    from metalign_amd.distributed import ShardJob
    gb, go = synth.make_genomes(40, 20000)
    rb, ro, src = synth.make_reads(gb, go, 60000, npresent=7)
    recs = synth.make_alignment_records(src + 1, 41)
    ref2tax = np.arange(41, dtype=np.uint32)
    k, n = 21, 200
    dbh, dbo = hip.sketch_genomes(gb, go, k, n)
    job = ShardJob(hip, None, 0, 1, k=k)
    job.load(rb, ro, recs, ref2tax, dbh, dbo)
    one = job.step(want_multimapped=True)
    piped = job.run(4, want_multimapped=True)  # single shard: every pass is queued before the previous one is read back
    again = job.step(want_multimapped=True)
    oh, oc, otr, _ = oracle_lib.sketch_reads(rb, ro, k, hmax=int(dbh.max()))
    nfiltered = util.job_sketch_size(oracle_lib, job, 0, rb, ro, k, dbh)  # the job's sketch
    ohits, osizes = oracle_lib.containment(oh, oc, otr, 2, dbh, dbo)
    want_c = oracle_lib.profile_assign(recs, ref2tax, 41, 0.5)
    for got in (one, piped, again):
        assert np.array_equal(got["hits"], ohits) and np.array_equal(got["sizes"], osizes)
        assert got["sketch_size"] == nfiltered < len(oh)
        for key in ("count", "bases", "first_seen"):
            assert np.array_equal(got[key], want_c[key]), key
        assert got["tot_rds"] == want_c["tot_rds"] and got["n_ambig"] == want_c["n_ambig"]
        off, tax, hl, rd = got["multimapped"]
        assert np.array_equal(off, want_c["mm_offsets"]) and np.array_equal(tax, want_c["mm_tax"])
        assert np.array_equal(hl, want_c["mm_hitlen"]) and np.array_equal(rd, want_c["mm_read"])


@pytest.mark.parametrize("flags", [[0, 16, 256, 272], [99, 147, 355, 403, 65, 129, 73, 137, 2048]])
def test_stage_c_reads_longer_than_a_tile(hip, oracle_lib, flags):
    """Reads with hundreds to thousands of alignment lines: they outrun the 64-descriptor halo that a tile stages
    past its end (the walk continues in HBM) and can span several 2048-record tiles; single-end and paired flags."""
    rng = np.random.default_rng(len(flags))
    nref, ntax = 300, 23
    ref2tax = rng.integers(0, ntax, size=nref).astype(np.uint32)
    lens = np.concatenate([rng.integers(1, 4, size=3000), rng.integers(60, 200, size=40), [2047, 2048, 2049, 5000, 70, 1]])
    rng.shuffle(lens)
    n = int(lens.sum())
    recs = np.zeros(n, dtype=oracle_lib.REC_DTYPE)
    starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
    new = np.zeros(n, dtype=np.uint32)
    new[starts] = 1
    # long reads hit few taxa (so that paired intersections are non-trivial), short ones anything
    ref = rng.integers(0, nref, size=n)
    recs["ref_new"] = ref.astype(np.uint32) | (new << 31)
    recs["total"] = 100
    recs["matched"] = rng.integers(20, 101, size=n)
    recs["flag_len"] = rng.choice(flags, size=n).astype(np.uint32) | (np.where(rng.random(n) < 0.8, 100, 0).astype(np.uint32) << 12)
    got = hip.profile_assign(recs, ref2tax, ntax, 0.5)
    want = oracle_lib.profile_assign(recs, ref2tax, ntax, 0.5)
    for key in want:
        assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), key
    assert len(want["mm_hitlen"]) > 50


def test_containment_dense_read_sketch_sparse_table(hip, oracle_lib):
    """A read sketch far denser than the table: the run that matches a tile of (hash, genome) pairs exceeds the LDS
    stage, so every pair goes through the bucket index on its own.  Exact against the oracle, ci = 1 and 2, with
    hashes shared between genomes."""
    rng = np.random.default_rng(11)
    top = 1 << 40
    pool = np.unique(rng.integers(1, top, size=120000, dtype=np.uint64))
    qh = pool[rng.random(len(pool)) < 0.6]                      # ~70 k entries in [1, 2^40)
    qc = rng.integers(1, 4, size=len(qh)).astype(np.uint32)
    genomes = []
    shared = rng.choice(pool, size=60, replace=False)
    for g in range(5):
        own = rng.choice(pool, size=240, replace=False)
        genomes.append(np.unique(np.concatenate([own, shared])))
    genomes.append(np.zeros(0, dtype=np.uint64))               # an empty genome sketch
    dbh = np.concatenate(genomes).astype(np.uint64)
    dbo = np.concatenate([[0], np.cumsum([len(g) for g in genomes])]).astype(np.uint64)
    d_h, d_c = hip.array(qh), hip.array(qc)
    sk = hip.sketch_from_pairs_dev(d_h.ptr, d_c.ptr, len(qh), 21)
    table = hip.upload_table(dbh, dbo)
    for ci in (1, 2):
        hits, sizes = hip.containment(sk, table, ci)
        ohits, osizes = oracle_lib.containment(qh, qc, False, ci, dbh, dbo)
        assert np.array_equal(hits, ohits) and np.array_equal(sizes, osizes)
        assert hits[:5].min() > 50 and hits[5] == 0 and sizes[5] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("k", [21, 32, 60])
def test_sketch_three_walks_of_a_tile(hip, oracle_lib, k):
    """k_sketch_reads picks one of three walks per tile of 64 reads: equally long reads without an invalid base, ragged
    reads without one (incl. reads shorter than k and empty ones), and tiles holding an invalid base.  Each against the
    oracle — hashes, counts and the number of k-mers seen — with both the table path and the list path."""
    rng = np.random.default_rng(100 + k)
    gb, go = util.random_genomes(rng, 5, 30000)
    hmax = int(0.05 * 2**64)
    uniform = util.sample_reads(rng, gb, go, 6000, 150, err=0.01)[:2]
    ragged_b, ragged_o, _ = util.sample_reads(rng, gb, go, 6000, 150, err=0.01, ragged=True, lower=True)
    lens = np.diff(ragged_o).astype(np.int64)
    lens[::97] = 0            # empty reads
    lens[5::101] = k - 1      # reads one base too short for a k-mer
    keep = np.concatenate([np.arange(int(ragged_o[i]), int(ragged_o[i]) + lens[i]) for i in range(len(lens))])
    ragged = (ragged_b[keep], np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64))
    dirty_b = uniform[0].copy()
    dirty_b[rng.integers(0, dirty_b.size, size=40)] = ord("N")  # a few tiles with an invalid base, most without
    dirty = (dirty_b, uniform[1])
    for name, (b, o) in (("uniform", uniform), ("ragged", ragged), ("dirty", dirty)):
        oh, oc, otr, oseen = oracle_lib.sketch_reads(b, o, k, hmax=hmax)
        d_b, d_o = hip.array(b if b.size else np.zeros(1, np.uint8)), hip.array(o)
        sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(o) - 1, k, hmax, 0)
        gh, gc = sk.download()
        assert np.array_equal(gh, oh) and np.array_equal(gc, oc), name
        assert sk.kmers_seen == oseen, (name, sk.kmers_seen, oseen)
        sk.free(); d_b.free(); d_o.free()
