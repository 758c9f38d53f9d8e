"""-m gpu: the 8-GPU configurations of BASELINE.json (configs[3], configs[4]) on ONE GPU.

A single box cannot run eight ranks, so the multi-rank path is covered three ways:
  * test_hash_range_sharding_emulated: the W ranks of metalign_amd/distributed.py run one after the other through
    the REAL kernels — per rank the multi-k read sketch of its read shard (through the full table's membership filter),
    cut at the hash-range bounds; per hash range the W slices merged by mg_sketch_merge_dev_async and run against
    table_slice(...); hits and sizes summed over the ranges; stage C sharded with carried state and lookahead —
    and must equal the UNSHARDED oracle.  W = 2 and W = 8, K = {21,31,51} (fused kernel) and a set without one.
  * test_config3_one_rank_share_full_size: one rank's share of configs[3] at full size — 12.5M reads against a 1/8
    hash-range slice of a 200k-genome x 1000-hash table (5 kb genomes: the dense regime) — through size-independent
    properties (sum over the eight ranges == unsharded) and against the oracle on a genome subsample.
  * test_config4_one_rank_stage_c_full_size: the stage-C shape of configs[4] per rank (12.5M records, 10 001 taxa),
    whole and as 8 shards, against the oracle.
The collectives themselves run under gloo at world 2 and 3 (tests/test_distributed_gloo.py) and under RCCL at world 1
(tests/dist_single_rank.py)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import util
from metalign_amd import formats, synth
from metalign_amd.distributed import table_bounds, table_max_hash, table_slice
from test_gpu_fullsize import _stage_c_sharded

pytestmark = pytest.mark.gpu


def _rank_slices(hip, d_b, d_o, nreads, ks, hmaxs, filts, bounds):
    """Stage A of one rank: -> per k (hashes, counts) of the rank's sketch and its cut points at the bounds."""
    sks = hip.sketch_reads_multi_dev_async(d_b.ptr, d_o.ptr, nreads, list(ks), hmaxs, 0, filts)
    out = []
    for ki, sk in enumerate(sks):
        h, c = sk.download()
        cuts = [0] + sk.split(bounds[ki][1:-1]) + [len(h)]
        out.append((h, c, cuts))
        sk.free()
    return out


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("ks", [(21, 31, 51), (25, 33)])
def test_hash_range_sharding_emulated(hip, oracle_lib, world, ks):
    rng = np.random.default_rng(world * 100 + len(ks))
    G, n = 300, 400
    gb, go = util.random_genomes(rng, G, 6000)
    tables = [oracle_lib.sketch_genomes(gb, go, k, n) for k in ks]
    rb, ro, src = util.sample_reads(rng, gb, go, 24000, 150, err=0.01, present=rng.choice(G, size=25, replace=False))
    nreads = len(ro) - 1
    hmaxs = [table_max_hash(h, o) for h, o in tables]
    bounds = [table_bounds(h, world, hm) for (h, _), hm in zip(tables, hmaxs)]  # equal-load ranges, as ShardJob.load cuts them
    filts = [hip.filter_build(h) for h, _ in tables]  # from the FULL table, as ShardJob.load does
    # ---- stage A per rank (contiguous read shards) ----
    rcuts = [nreads * r // world for r in range(world + 1)]
    per_rank = []
    for r in range(world):
        b0, b1 = int(ro[rcuts[r]]), int(ro[rcuts[r + 1]])
        d_b = hip.array(rb[b0:b1] if b1 > b0 else np.zeros(1, np.uint8))
        d_o = hip.array(ro[rcuts[r]: rcuts[r + 1] + 1] - ro[rcuts[r]])
        per_rank.append(_rank_slices(hip, d_b, d_o, rcuts[r + 1] - rcuts[r], ks, hmaxs, filts, bounds))
    # ---- per hash range: the all-to-all's delivery, the merge, stage B against the table slice ----
    for ki, k in enumerate(ks):
        dbh, dbo = tables[ki]
        hits = np.zeros(G, dtype=np.uint64)
        sizes = np.zeros(G, dtype=np.uint64)
        total_entries = 0
        for q in range(world):
            lo, hi = bounds[ki][q], bounds[ki][q + 1]
            rh = np.concatenate([per_rank[r][ki][0][per_rank[r][ki][2][q]: per_rank[r][ki][2][q + 1]] for r in range(world)])
            rc = np.concatenate([per_rank[r][ki][1][per_rank[r][ki][2][q]: per_rank[r][ki][2][q + 1]] for r in range(world)])
            assert rh.size == 0 or (int(rh.min()) >= lo and int(rh.max()) < hi)
            d_h, d_c = hip.array(rh if rh.size else np.zeros(1, np.uint64)), hip.array(rc if rc.size else np.zeros(1, np.uint32))
            merged = hip.sketch_merge_dev_async(d_h.ptr, d_c.ptr, rh.size, k, lo, hi - 1)
            sh, so = table_slice(dbh, dbo, lo, hi)
            table = hip.upload_table(sh, so)
            hq, sq = hip.containment(merged, table, 2)
            merged.resolve()
            total_entries += merged.size
            hits += hq
            sizes += sq
            table.free()
            merged.free()
        qh, qc, tr, _ = oracle_lib.sketch_reads_filtered(rb, ro, k, dbh, hmax=hmaxs[ki])
        ohits, osizes = oracle_lib.containment(qh, qc, tr, 2, dbh, dbo)
        assert np.array_equal(hits, ohits) and np.array_equal(sizes, osizes), (k, world)
        assert total_entries == len(qh)
        assert ohits.max() > 0.5 * n
    # ---- stage C: W shards with carried state and lookahead == the whole stream ----
    recs = synth.make_alignment_records(src + 1, G + 1)
    ref2tax = rng.integers(0, 37, size=G + 1).astype(np.uint32)
    want = oracle_lib.profile_assign(recs, ref2tax, 37, 0.5)
    starts = np.nonzero(recs["ref_new"] >> 31)[0]
    cuts = [int(starts[len(starts) * r // world]) for r in range(1, world)]
    got = _stage_c_sharded(hip, recs, ref2tax, 37, cuts)
    for key in want:
        assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), (key, world)


def _oracle_sketch_threads(oracle_lib, rb, ro, k, hmax, table_hashes, nthreads):
    """The filtered read sketch of all reads, computed in contiguous shares on `nthreads` threads and merged."""
    nreads = len(ro) - 1
    bits, mask = oracle_lib.filter_bits(table_hashes)
    cuts = [nreads * i // nthreads for i in range(nthreads + 1)]

    def share(i):
        lo, hi = cuts[i], cuts[i + 1]
        b0, b1 = int(ro[lo]), int(ro[hi])
        h, c, _, _ = oracle_lib.sketch_reads(rb[b0:b1], ro[lo: hi + 1] - ro[lo], k, hmax=hmax)
        keep = bits[(h & mask).astype(np.int64)]
        return h[keep], c[keep]

    with ThreadPoolExecutor(nthreads) as ex:
        parts = list(ex.map(share, range(nthreads)))
    allh = np.concatenate([p[0] for p in parts])
    allc = np.concatenate([p[1] for p in parts]).astype(np.uint64)
    uh, inv = np.unique(allh, return_inverse=True)
    uc = np.minimum(np.bincount(inv, weights=allc, minlength=len(uh)), oracle_lib.DEFAULT_CS).astype(np.uint32)
    return uh, uc


@pytest.mark.skipif(os.environ.get("MG_TEST_CONFIG3_HASH") != "1",
                    reason="the hash-range path of rounds 2-5 at configs[3] size (68 s): on request, MG_TEST_CONFIG3_HASH=1; since round 6 "
                           "bench.py --gpus N runs stage A by k-mer identity (test_config3_default_path_at_full_size_with_every_collective)")
def test_config3_one_rank_share_full_size(hip, oracle_lib):
    """BASELINE.json configs[3] as ONE of its 8 ranks sees it: 12.5M reads (100M / 8) and a 1/8 hash-range slice of the
    200k-genome table, K = {21,31,51}; 5 kb genomes put the table's largest hash at a fifth of the hash range — the
    dense regime, where the membership filter and the counting table carry the load."""
    W, rank, ks, n = 8, 3, (21, 31, 51), 1000
    G, glen, nreads = 200_000, 5_000, 12_500_000
    gb, go = synth.make_genomes(G, glen)
    rb, ro, src = synth.make_reads(gb, go, nreads, npresent=G // 20, seed=synth.SEED + 1 + 1000 * rank)
    d_b, d_o = hip.array(rb), hip.array(ro)
    depth = np.bincount(src, minlength=G) * 150 / glen
    nthreads = max(1, min(os.cpu_count() or 1, 64))
    sub = np.sort(np.random.default_rng(3).choice(G, size=64, replace=False))
    for ki, k in enumerate(ks):
        dbh, dbo = hip.sketch_genomes(gb, go, k, n)
        assert len(dbh) == G * n
        hmax = table_max_hash(dbh, dbo)
        assert 0.1 < hmax / 2.0 ** 64 < 0.4  # dense: the threshold alone filters little
        filt = hip.filter_build(dbh)
        sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, k, hmax, 0, filt=filt)
        h, c = sk.download()
        assert np.all(h[1:] > h[:-1]) and int(c.max()) <= 3 and int(h[-1]) <= hmax
        # unsharded stage B on this rank's reads
        full = hip.upload_table(dbh, dbo)
        hits_full, sizes_full = hip.containment(sk, full, 2)
        full.free()
        assert np.all(sizes_full == n)
        ci = hits_full / n
        assert ci[depth > 30].min() > 0.8 and ci[depth == 0].max() < 0.05
        # the rank's own range — and, for the middle k, every range: the sum over the ranges is the unsharded result
        b = table_bounds(dbh, W, hmax)
        cuts = [0] + sk.split(b[1:W]) + [len(h)]
        ranges = range(W) if ki == 1 else [rank]
        hits_sum, sizes_sum = np.zeros(G, np.uint64), np.zeros(G, np.uint64)
        for q in ranges:
            sh, so = table_slice(dbh, dbo, b[q], b[q + 1])
            assert abs(len(sh) - G * n / W) < 0.001 * G * n / W  # quantile bounds: the slices are equal-sized
            hq, cq = h[cuts[q]: cuts[q + 1]], c[cuts[q]: cuts[q + 1]]
            d_h, d_c = hip.array(hq), hip.array(cq)
            merged = hip.sketch_merge_dev_async(d_h.ptr, d_c.ptr, hq.size, k, b[q], b[q + 1] - 1)  # (one source: itself)
            table = hip.upload_table(sh, so)
            hit_q, size_q = hip.containment(merged, table, 2)
            assert not merged.resolve() and merged.size == hq.size
            mh, mc = merged.download()
            assert np.array_equal(mh, hq) and np.array_equal(mc, cq)
            hits_sum += hit_q
            sizes_sum += size_q
            table.free()
            merged.free()
            d_h.free()
            d_c.free()
        if ki == 1:
            assert np.array_equal(hits_sum, hits_full) and np.array_equal(sizes_sum, sizes_full)
        else:
            assert np.all(hits_sum <= hits_full) and abs(int(sizes_sum.sum()) - G * n // W) < 0.001 * G * n / W
        # the oracle (threads over read shares) on a genome subsample, against the full-size GPU result
        uh, uc = _oracle_sketch_threads(oracle_lib, rb, ro, k, hmax, dbh, nthreads)
        assert np.array_equal(uh, h) and np.array_equal(uc, c)
        sub_h = np.concatenate([dbh[int(dbo[g]):int(dbo[g + 1])] for g in sub])
        sub_o = np.arange(len(sub) + 1, dtype=np.uint64) * np.uint64(n)
        ohits, osizes = oracle_lib.containment(uh, uc, False, 2, sub_h, sub_o)
        assert np.array_equal(ohits, hits_full[sub]) and np.array_equal(osizes, sizes_full[sub])
        sk.free()
        filt.free()


def test_config4_one_rank_stage_c_full_size(hip, oracle_lib):
    """configs[4] per rank: 10M reads' worth of alignment records (12.5M) against 10 001 taxa — whole, and as the eight
    shards a node would cut it into — equal to the oracle in every accumulator and every multimapped list."""
    nreads, T = 10_000_000, 10_001
    rng = np.random.default_rng(44)
    src = rng.integers(0, T - 1, size=nreads)  # every taxon is hit: the per-workgroup bins see all 10 001
    recs = synth.make_alignment_records(src + 1, T)
    ref2tax = np.arange(T, dtype=np.uint32)
    want = oracle_lib.profile_assign(recs, ref2tax, T, 0.5)
    got = hip.profile_assign(recs, ref2tax, T, 0.5)
    for key in want:
        assert np.array_equal(np.asarray(got[key]), np.asarray(want[key])), key
    starts = np.nonzero(recs["ref_new"] >> 31)[0]
    cuts = [int(starts[len(starts) * r // 8]) for r in range(1, 8)]
    sharded = _stage_c_sharded(hip, recs, ref2tax, T, cuts)
    for key in want:
        assert np.array_equal(np.asarray(sharded[key]), np.asarray(want[key])), key


def test_hash_major_table_on_disk_sliced_loads(hip, oracle_lib, tmp_path):
    """The table as the builder writes it (formats version 2: hash-major pairs + the membership filter): uploaded without
    a sort (mg_db_upload_sorted) it gives the containment of the genome-major upload; a rank of a W-rank job maps only
    its hash range [bounds[r], bounds[r+1]) — 1/W of the pair files — and the per-range counts add up to the whole."""
    rng = np.random.default_rng(12)
    G, n, ks, W = 200, 300, (21, 31), 4
    gb, go = util.random_genomes(rng, G, 5000)
    rb, ro, _ = util.sample_reads(rng, gb, go, 8000, 150, err=0.01, present=rng.choice(G, size=12, replace=False))
    d_b, d_o = hip.array(rb), hip.array(ro)
    per_k, filters = {}, {}
    for k in ks:
        per_k[k] = hip.sketch_genomes(gb, go, k, n)
        f = hip.filter_build(per_k[k][0])
        filters[k] = f.download()
        f.free()
    formats.write_sketch_table(str(tmp_path / "t"), ["g%d" % g for g in range(G)], list(ks), n, per_k, filters)
    disk = formats.SketchTable(str(tmp_path / "t"))
    for k in ks:
        dbh, dbo = per_k[k]
        full = disk.pairs(k)
        filt = hip.filter_from_bits(disk.filter_bits(k))
        assert np.array_equal(filt.download(), filters[k])
        sk = hip.sketch_reads_dev(d_b.ptr, d_o.ptr, len(ro) - 1, k, full["max_hash"], 0, filt=filt)
        want_h, want_s = hip.containment(sk, hip.upload_table(dbh, dbo), 2)   # genome-major upload: sorted on the device
        got_h, got_s = hip.containment(sk, hip.upload_table_sorted(**full), 2)
        assert np.array_equal(got_h, want_h) and np.array_equal(got_s, want_s)
        qh, qc, tr, _ = oracle_lib.sketch_reads_filtered(rb, ro, k, dbh, hmax=full["max_hash"])
        oh, osz = oracle_lib.containment(qh, qc, tr, 2, dbh, dbo)
        assert np.array_equal(got_h, oh) and np.array_equal(got_s, osz)
        b = table_bounds(full["pair_hash"], W, full["max_hash"], presorted=True)
        assert b == table_bounds(dbh, W, full["max_hash"])  # the same cut points from either layout
        hits, sizes, touched = np.zeros(G, np.uint64), np.zeros(G, np.uint64), 0
        for r in range(W):
            part = disk.pairs(k, b[r], b[r + 1])
            touched += len(part["pair_hash"])
            assert abs(len(part["pair_hash"]) - len(dbh) / W) <= 0.02 * len(dbh) / W + G  # 1/W of the table per rank
            hr, sr = hip.containment(sk, hip.upload_table_sorted(**part), 2)
            hits += hr
            sizes += sr
        assert touched == len(dbh) and np.array_equal(hits, oh) and np.array_equal(sizes, osz)


# (rounds 1-3's sketch per k at this size costs the ORACLE three minutes per sample — three 60M-hash sketches merged, 6 x 10^8
# table hashes walked — so it runs on request only, MG_TEST_CONFIG3_SKETCH_PER_K=1; its 2M-read result of round 4, forced
# overflow included, is recorded in profiles/r04/config3_checks.txt)
# (round 6: what `bench.py --gpus N` runs is stage A by k-mer identity — every rank the whole 200k-genome table and its k-mer index, the ranks'
# counters all-gathered; the hash-range path of rounds 2-5 at this size on request, MG_TEST_CONFIG3_HASH=1: its results are in
# profiles/r04/config3_checks.txt and it is unchanged)
_C3 = [("reference_pipeline", 0, "kmer")] + ([("reference_pipeline", 0, "hash")] if __import__("os").environ.get("MG_TEST_CONFIG3_HASH") == "1" else []) + \
    ([("sketch_per_k", 0, None)] if __import__("os").environ.get("MG_TEST_CONFIG3_SKETCH_PER_K") == "1" else [])


@pytest.mark.parametrize("definition,mode,match", _C3)
def test_config3_default_path_at_full_size_with_every_collective(definition, mode, match):
    """bench.py --config 3 (the N > 1 driver workload) AS THE JOB RUNS IT on a rank, at full size: 12.5M reads, 15.6M
    records, the 200k-genome table with the resident index the job chooses for itself, RCCL at world size 1 with every
    collective of the pass in the path — a >= 2M-read sample against the threaded C oracle (hits and sizes of all 200 000
    genomes for every k, every stage-C accumulator), the same with every list / table undersized (overflow -> redo at this
    size), then step() == run(3) on the whole workload.  tests/dist_config3_full.py, a child process (torch.distributed
    stays out of this one)."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29640 + mode + (2 if definition == "sketch_per_k" else 0)))
    # (the default definition — the reference pipeline — on >= 2M reads, with and without the forced overflow: 36 s; rounds
    # 1-3's sketch per k, whose oracle merges three 60M-hash sketches and walks 6 x 10^8 table hashes per sample, on 500k
    # reads without the repeat: the full-size figures of that path are in profiles/r04/config3_checks.txt, 2M reads, 456 s)
    extra = ["2000000", "plain", "kmer"] if match == "kmer" else ([] if definition == "reference_pipeline" else ["500000", "plain"])
    r = subprocess.run([sys.executable, os.path.join(here, "dist_config3_full.py"), definition, str(mode)] + extra, capture_output=True,
                       text=True, timeout=1500, env=env)
    assert r.returncode == 0 and "config3-full ok" in r.stdout, (r.stdout[-3000:], r.stderr[-6000:])
    rep = json.loads(r.stdout.split("config3-full ok ", 1)[1].splitlines()[0])
    if match == "kmer":
        assert rep["check_hint_None"]["sample_reads"] >= 2_000_000 and rep["match"] == "kmer" and rep["resident_index_bytes"] == 0
    elif definition == "reference_pipeline":
        assert rep["check_hint_None"]["sample_reads"] >= 2_000_000 and rep["check_hint_0.002"]["sample_reads"] >= 2_000_000
        assert rep["resident_index_bytes"] > 0  # (the job chose the index for this dense table, and kept it after measuring)
    print(json.dumps(rep))
