"""-m gpu: BASELINE.json configurations at FULL size, checked through size-independent properties, plus the
sharded C-ABI of stage C (begin / state_map / commit with lookahead) on the real GPU."""
import numpy as np
import pytest

from metalign_amd import _hip, synth
from metalign_amd.distributed import compose_incoming

pytestmark = pytest.mark.gpu
U64_MAX = 0xFFFFFFFFFFFFFFFF
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    _COMP[a] = b


def _stage_c_sharded(hip, recs, ref2tax, ntax, cuts, pct_id=0.5):
    """Run stage C as len(cuts)+1 shards through mg_profile_begin_dev / _commit_dev; -> same dict as profile_assign."""
    bounds = [0] + list(cuts) + [len(recs)]
    d_r2t = hip.array(ref2tax)
    acc = hip.empty(3 * ntax + 2, np.uint64)
    host = np.zeros(3 * ntax + 2, dtype=np.uint64)
    host[2 * ntax:3 * ntax] = U64_MAX
    acc.upload(host)
    shards, arrays = [], []
    for i in range(len(bounds) - 1):
        a, b = bounds[i], bounds[i + 1]
        look = i + 1 < len(bounds) - 1 or False
        has_look = b < len(recs)
        part = recs[a:b + (1 if has_look else 0)]
        d = hip.array(part if len(part) else np.zeros(1, _hip.REC_DTYPE))
        arrays.append(d)
        shards.append(hip.profile_begin_dev(d.ptr, b - a, has_look, d_r2t.ptr, len(ref2tax), ntax, pct_id))
    maps = [s.state_map() for s in shards]
    groups = [s.ngroups for s in shards]
    mm_all = []
    base = acc.ptr
    first_nonempty = next(i for i in range(len(shards)) if bounds[i + 1] > bounds[i])
    for i, s in enumerate(shards):
        s.commit(compose_incoming(maps, i), i == first_nonempty, sum(groups[:i]), base, base + 8 * ntax,
                 base + 16 * ntax, base + 24 * ntax)
        mm_all.append(s.multimapped())
    out = acc.download()
    off = [np.zeros(1, np.uint64)]
    tot_e = 0
    for o, t, h, r in mm_all:
        off.append(o[1:] + np.uint64(tot_e))
        tot_e += len(t)
    res = dict(count=out[:ntax], bases=out[ntax:2 * ntax], first_seen=out[2 * ntax:3 * ntax], tot_rds=int(out[3 * ntax]),
               n_ambig=int(out[3 * ntax + 1]), mm_offsets=np.concatenate(off),
               mm_tax=np.concatenate([m[1] for m in mm_all]), mm_hitlen=np.concatenate([m[2] for m in mm_all]),
               mm_read=np.concatenate([m[3] for m in mm_all]))
    for s in shards:
        s.free()
    return res


def _random_records(rng, n, nref, p_new, flags, oracle_dtype):
    recs = np.zeros(n, dtype=oracle_dtype)
    new = rng.random(n) < p_new
    new[0] = True
    recs["ref_new"] = rng.integers(0, nref, size=n).astype(np.uint32) | (new.astype(np.uint32) << 31)
    recs["total"] = 100
    recs["matched"] = rng.integers(20, 101, size=n)
    recs["flag_len"] = rng.choice(flags, size=n).astype(np.uint32) | (np.where(rng.random(n) < 0.8, 100, 0).astype(np.uint32) << 12)
    return recs, np.nonzero(new)[0]


@pytest.mark.parametrize("p_new,flags", [(0.7, [0, 16, 256, 272]), (0.95, [0, 16]), (0.5, [99, 147, 355, 403, 65, 129, 73, 137, 2048]),
                                         (1.0, [0])])
def test_stage_c_shards_on_gpu_equal_whole_stream(hip, oracle_lib, p_new, flags):
    """Carried state across shard edges on the GPU: every split of the stream gives the unsharded oracle result.
    p_new = 1.0 is the all-single-line stream: one identity cascade from the first record to the last."""
    rng = np.random.default_rng(int(p_new * 100))
    n, nref, ntax = 200000, 50, 11
    ref2tax = rng.integers(0, ntax, size=nref).astype(np.uint32)
    recs, starts = _random_records(rng, n, nref, p_new, flags, oracle_lib.REC_DTYPE)
    want = oracle_lib.profile_assign(recs, ref2tax, ntax, 0.5)
    for cuts in ([int(starts[len(starts) // 2])], [int(starts[3]), int(starts[len(starts) // 3]), int(starts[-2])],
                 [int(starts[1]), int(starts[1]), int(starts[2])]):  # an empty middle shard too
        got = _stage_c_sharded(hip, recs, ref2tax, ntax, sorted(cuts))
        for key in want:
            assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), (key, cuts)


def _containment_for(hip, d_b, d_o, nreads, k, dbh, dbo):
    table = hip.upload_table(dbh, dbo)
    sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, k, table.max_hash, 0)
    hits, sizes = hip.containment(sk, table, 2)
    h, c = sk.download()
    return hits, sizes, h, c


def test_config1_full_size_properties(hip):
    """BASELINE.json configs[1]: 1M synthetic 150 bp reads vs a 1k-genome sketch DB, k = 21."""
    k, n = 21, 1000
    gb, go = synth.make_genomes(1000, 50_000)
    rb, ro, src = synth.make_reads(gb, go, 1_000_000)
    dbh, dbo = hip.sketch_genomes(gb, go, k, n)
    assert np.all(np.diff(dbo.astype(np.int64)) == n)
    per = dbh.reshape(1000, n)
    assert np.all(per[:, 1:] > per[:, :-1])  # every genome sketch strictly ascending (sorted, distinct)
    d_b, d_o = hip.array(rb), hip.array(ro)
    hits, sizes, h, c = _containment_for(hip, d_b, d_o, len(ro) - 1, k, dbh, dbo)
    assert np.all(h[1:] > h[:-1]) and h[-1] <= dbh.max()
    assert int(c.astype(np.uint64).sum()) <= 130 * 1_000_000
    present = np.unique(src)
    ci = hits / np.maximum(sizes, 1)
    covered = np.bincount(src, minlength=1000) * 150 / 50_000 > 8  # >= 8x coverage: nearly every sketch k-mer seen twice
    assert ci[covered].min() > 0.9
    absent = np.setdiff1d(np.arange(1000), present)
    assert ci[absent].max() < 0.02
    # strand invariance: reverse-complementing every read leaves the sketch unchanged
    rc = _COMP[rb.reshape(-1, 150)[:, ::-1]].reshape(-1)
    d_rc = hip.array(rc)
    sk2 = hip.sketch_reads_dev(d_rc.ptr, d_o.ptr, len(ro) - 1, k, int(dbh.max()), 0)
    h2, c2 = sk2.download()
    assert np.array_equal(h, h2) and np.array_equal(c, c2)
    # two read shards merged == one pass
    half = 500_000
    cut = int(ro[half])
    parts = [(rb[:cut], ro[: half + 1]), (rb[cut:], ro[half:] - ro[half])]
    hs, cs = [], []
    for b, o in parts:
        hh, cc, t, _ = hip.sketch_reads(b, o, k, hmax=int(dbh.max()))
        hs.append(hh); cs.append(cc)
    allh, allc = np.concatenate(hs), np.concatenate(cs)
    d_h, d_c = hip.array(allh), hip.array(allc)  # keep the device arrays alive across the call
    merged = hip.sketch_from_pairs_dev(d_h.ptr, d_c.ptr, allh.size, k)
    mh, mc = merged.download()
    assert np.array_equal(mh, h) and np.array_equal(mc, c)
    # stage C conservation: every processed read is exactly one of unique / multimapped / Ambiguous
    recs = synth.make_alignment_records(src + 1, 1001)
    res = hip.profile_assign(recs, np.arange(1001, dtype=np.uint32), 1001, 0.5)
    assert res["tot_rds"] == 1_000_000
    assert int(res["count"].sum()) + len(res["mm_hitlen"]) + (res["n_ambig"] - 1) == res["tot_rds"] - 1
    # a unique read's bases = SEQ lengths of ALL its lines (secondaries carry '*': 0), so <= 150 per read
    assert 0 < int(res["bases"].sum()) <= 150 * int(res["count"].sum())
    assert np.all(np.diff(res["mm_read"].astype(np.int64)) > 0)


def test_config2_full_size_properties(hip):
    """BASELINE.json configs[2]: 10M reads vs a 10k-genome DB, multi-k {21,31,51} containment."""
    ks, n = (21, 31, 51), 1000
    gb, go = synth.make_genomes(10_000, 20_000)
    rb, ro, src = synth.make_reads(gb, go, 10_000_000, npresent=200)
    d_b, d_o = hip.array(rb), hip.array(ro)
    depth = np.bincount(src, minlength=10_000) * 150 / 20_000
    present = depth > 0
    prev = None
    for k in ks:
        dbh, dbo = hip.sketch_genomes(gb, go, k, n)
        hits, sizes, h, c = _containment_for(hip, d_b, d_o, len(ro) - 1, k, dbh, dbo)
        ci = hits / np.maximum(sizes, 1)
        assert np.all(h[1:] > h[:-1])
        assert ci[depth > 12].min() > 0.85 and ci[~present].max() < 0.02
        if prev is not None:  # 1 % substitution errors cost longer k-mers more: containment does not grow with k
            assert np.mean(ci[present]) <= np.mean(prev[present]) + 1e-3
        prev = ci
    recs = synth.make_alignment_records(src + 1, 10_001)
    ref2tax = np.arange(10_001, dtype=np.uint32)  # > 2048 taxa: global-atomic histogram path
    res = hip.profile_assign(recs, ref2tax, 10_001, 0.5)
    assert res["tot_rds"] == 10_000_000
    assert int(res["count"].sum()) + len(res["mm_hitlen"]) + (res["n_ambig"] - 1) == res["tot_rds"] - 1
    starts = np.nonzero(recs["ref_new"] >> 31)[0]
    cuts = [int(starts[len(starts) // 4]), int(starts[len(starts) // 2]), int(starts[3 * len(starts) // 4])]
    sharded = _stage_c_sharded(hip, recs, ref2tax, 10_001, cuts)
    for key in ("count", "bases", "first_seen", "tot_rds", "n_ambig", "mm_read", "mm_tax", "mm_hitlen", "mm_offsets"):
        assert np.array_equal(np.asarray(sharded[key]), np.asarray(res[key])), key


@pytest.mark.parametrize("definition,mode,match", [("reference_pipeline", 0, "kmer"), ("reference_pipeline", 0, "hash"),
                                                   ("reference_pipeline", 1, "kmer"), ("sketch_per_k", 0, None)])
def test_the_benchmarked_workload_against_the_oracle(hip, oracle_lib, definition, mode, match):
    """BASELINE.json configs[2] EXACTLY as bench.py runs it at N = 1 (bench.build_workload: 10M reads of 500 present 50 kb
    genomes among 10 000, K = {21,31,51}, 12.5M alignment records, 10 001 taxa), under the headline's definition of stage
    A/B (the reference pipeline, k-mers met by identity — match "kmer", against the oracle's mgo_kmer_table_* — or by hash value)
    and the others bench.py reports beside it: ShardJob.step() — stage A, stage B
    with one column per k, stage C — on a >= 2M-read sample against the C oracle on every host core: hits and sizes of
    all 10 000 genomes for every k, count / bases / first_seen of every taxon, tot_rds, n_ambig.  (bench.py prints the
    same comparison as `check.oracle_equal`; here it is the driver's GPU test tier that holds it.)"""
    import argparse
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cfg = dict(bench.PRESETS[2], config=2, custom=False, match=match, definition=definition, hash_mode=mode)
    try:
        w = bench.build_workload(cfg, 1000, 0, hip, definition, mode)
        assert len(w["ro"]) - 1 == 10_000_000 and w["ntax"] == 10_001 and w["table_hashes"] == (1 if definition == "reference_pipeline" else 3) * 10_000_000
        # (the identity oracle looks every read k-mer up in a sorted table of ten million: 5 x 10^5 reads/s on 64 threads — a 4M-read sample)
        args = argparse.Namespace(cpu_seconds=8.0 if match == "kmer" else 20.0)
        base, check = bench.cpu_baseline_and_check(args, cfg, w, hip)
        nsample = int(check["compared"].split("sample (")[1].split(" reads")[0])
        assert nsample >= 2_000_000 or nsample == 10_000_000, check["compared"]
        assert check["oracle_equal"], check["mismatch"]
        assert (check["definition"], check["hash_mode"], check["match"]) == (definition, mode, match)
        # and the pipelined passes the benchmark times give the same sketches as single steps
        job = bench.make_job(hip, None, 0, 1, cfg, w)
        assert job.match == match
        one = job.step()
        out = job.run(3)
        assert out["sketch_sizes"] == one["sketch_sizes"] and out["tot_rds"] == one["tot_rds"]
        assert np.array_equal(out["hits_k"], one["hits_k"]) and out["hits_k"].shape == (3, 10_000)
        assert out["sketched_ks"] == ([51] if definition == "reference_pipeline" else [21, 31, 51])
        del job
    finally:
        hip.set_hash_mode(0)
        oracle_lib.set_hash_mode(0)


@pytest.mark.parametrize("mode", [0, 1])
def test_the_reference_s_own_parameters_at_the_benchmarked_size(hip, oracle_lib, mode):
    """`bench.py --preset stock`: the parameters the reference itself runs (scripts/select_db.py:44,50,69-70,75 — kmc -k60, the streaming
    query's range 30-60-10, n = 1000) on configs[2]'s sizes, both hash definitions: a 12-byte tail in MurmurHash3 and four prefix columns,
    which the headline's {21,31,51} does not exercise.  A >= 2M-read sample against the C oracle: hits and sizes of all 10 000 genomes for
    k = 30, 40, 50, 60, every per-taxon accumulator."""
    import argparse
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cfg = dict(bench.PRESETS[2], config=2, custom=False, ks=list(bench.STOCK_KS), name=bench.STOCK_NAME, match="kmer" if mode == 0 else "hash",
               definition="reference_pipeline", hash_mode=mode)
    try:
        w = bench.build_workload(cfg, 1000, 0, hip, "reference_pipeline", mode)
        assert len(w["ro"]) - 1 == 10_000_000 and w["table_hashes"] == 10_000_000
        args = argparse.Namespace(cpu_seconds=8.0 if mode == 0 else 20.0)
        base, check = bench.cpu_baseline_and_check(args, cfg, w, hip)
        nsample = int(check["compared"].split("sample (")[1].split(" reads")[0])
        assert nsample >= 2_000_000 or nsample == 10_000_000, check["compared"]
        assert check["oracle_equal"], check["mismatch"]
        job = bench.make_job(hip, None, 0, 1, cfg, w)
        assert job.match == cfg["match"] == check["match"]
        one = job.step()
        out = job.run(3)
        assert out["sketch_sizes"] == one["sketch_sizes"] and np.array_equal(out["hits_k"], one["hits_k"]) and out["hits_k"].shape == (4, 10_000)
        assert out["sketched_ks"] == [60]
        del job
    finally:
        hip.set_hash_mode(0)
        oracle_lib.set_hash_mode(0)


def test_reference_pipeline_table_at_the_benchmarked_size(hip, oracle_lib):
    """The table bench.py's headline runs against — 10 000 genomes x 1000 sketched 51-mers, prefix columns for k = 21 and 31 —
    as the device builder lays it out (mg_sketch_genomes_kmers + mg_refdb_build) against the oracle's, entry for entry: the
    sketches and their kept k-mers on a subset of the genomes (the oracle hashes every position on one core), the derived
    structures on all 10M pairs."""
    ks = [21, 31, 51]
    gb, go = synth.make_genomes(10_000, 50_000)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, 51, 1000)
    assert len(h) == 10_000_000 and np.all(np.diff(o.astype(np.int64)) == 1000)
    sub = [0, 1, 4999, 9999]
    for g in sub:
        a, b = int(go[g]), int(go[g + 1])
        oh, ohi, olo, oo = oracle_lib.sketch_genomes_kmers(gb[a:b], np.asarray([0, b - a], dtype=np.uint64), 51, 1000)
        sl = slice(int(o[g]), int(o[g + 1]))
        assert np.array_equal(h[sl], oh) and np.array_equal(khi[sl], ohi) and np.array_equal(klo[sl], olo), g
    table = hip.refdb_build(h, khi, klo, o, ks)
    got = table.download()
    table.free()
    want = oracle_lib.refpipe_build(h, khi, klo, o, ks)
    assert np.array_equal(got["pair_hash"], want["pair_hash"]) and np.array_equal(got["pair_gen"], want["pair_gen"])
    assert np.array_equal(got["kmer_hi"], want["kmer_hi"]) and np.array_equal(got["kmer_lo"], want["kmer_lo"])
    for k in ks[:-1]:
        assert got["small"][k]["nprefix"] == want["small"][k]["nprefix"]
        for key in ("pa", "pb", "cid", "cgen", "gsize"):
            assert np.array_equal(got["small"][k][key], want["small"][k][key]), (k, key)
