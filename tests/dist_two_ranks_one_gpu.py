"""Helper of tests/test_pipeline_gpu.py: TWO ranks (torch.distributed.run --nproc-per-node 2) sharing ONE GPU, every
kernel of the multi-GPU path real (stage A per read shard, slicing by hash range, the exchange of slices, the merge of
slices from two different ranks, stage B on a table slice, stage C with the carried state across the shard edge), the
collectives over gloo with the tensors staged through the host (RCCL refuses two ranks on one device).  Plain steps and
the four-passes-in-flight schedule, single k and the fused multi-k launch, against the oracle on the unsharded input."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import oracle  # noqa: E402
import util  # noqa: E402
from metalign_amd import synth  # noqa: E402
from metalign_amd._hip import Hip, debug_set  # noqa: E402
from metalign_amd.distributed import ShardJob  # noqa: E402

for _kv in filter(None, os.environ.get("MG_TEST_KNOBS", "").split(",")):  # (the parent test's knobs: the library reads no environment)
    debug_set(_kv.split("=")[0], int(_kv.split("=")[1]))

torch.cuda.set_device(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
hip = Hip.get(0, stream=stream.cuda_stream)
oracle.build()
G = 60
gb, go = synth.make_genomes(G, 20000)
rb, ro, src = synth.make_reads(gb, go, 80000, npresent=9)
recs = synth.make_alignment_records(src + 1, G + 1)
ref2tax = np.arange(G + 1, dtype=np.uint32)
want = oracle.profile_assign(recs, ref2tax, G + 1, 0.5)
# shards: reads and records in `world` contiguous parts (records cut on read boundaries)
nreads = len(ro) - 1
ncuts = [nreads * i // world for i in range(world + 1)]
starts = np.nonzero(recs["ref_new"] >> 31)[0]
rcuts = [0] + [int(starts[len(starts) * i // world]) for i in range(1, world)] + [len(recs)]
my_rb = rb[int(ro[ncuts[rank]]): int(ro[ncuts[rank + 1]])]
my_ro = ro[ncuts[rank]: ncuts[rank + 1] + 1] - ro[ncuts[rank]]
my_recs = recs[rcuts[rank]: rcuts[rank + 1]]
# (k set, s): one k given bare, the fused multi-k launch, and bottom-s sketches (the sample-wide cut over the ranks' slices)
for kspec, s_cut in ((21, 0), ([21, 31, 51], 0), (21, 500), ([21, 31, 51], 700)):
    ks = [kspec] if np.isscalar(kspec) else kspec
    # (MG_TEST_KNOBS distinct_hint_ppm: every counting table undersized -> sketches redone, words stale, the all-gather repeated)
    tabs = [hip.sketch_genomes(gb, go, k, 1000 if ("distinct_hint_ppm" in os.environ.get("MG_TEST_KNOBS", "")) else 200) for k in ks]
    job = ShardJob(hip, dist, rank, world, k=kspec, s=s_cut)
    if np.isscalar(kspec):
        job.load(my_rb, my_ro, my_recs, ref2tax, tabs[0][0], tabs[0][1])
    else:
        job.load(my_rb, my_ro, my_recs, ref2tax, [t[0] for t in tabs], [t[1] for t in tabs])
    outs = [job.step(want_multimapped=True), job.run(4, want_multimapped=True), job.step(want_multimapped=True)]
    for idx, got in enumerate(outs):
        for ki, k in enumerate(ks):
            dbh, dbo = tabs[ki]
            fh, fc, ftr, _ = oracle.sketch_reads_filtered(rb, ro, k, dbh, hmax=int(dbh.max()), s=s_cut)  # the job's sketch
            ohits, osizes = oracle.containment(fh, fc, ftr, 2, dbh, dbo)
            assert np.array_equal(got["hits_k"][ki], ohits) and np.array_equal(got["sizes_k"][ki], osizes), (rank, idx, k, s_cut)
            nsk = util.job_sketch_size(oracle, job, ki, rb, ro, k, dbh, s_cut)  # (the table's filter, or its resident index)
            assert got["sketch_sizes"][ki] == nsk, (rank, k, s_cut, got["sketch_sizes"][ki], nsk, len(fh))
            assert not s_cut or (ftr and len(fh) == s_cut), (k, s_cut, len(fh))  # (the cut is exercised)
        for key in ("count", "bases", "first_seen"):
            assert np.array_equal(got[key], want[key]), (rank, idx, key)
        assert got["tot_rds"] == want["tot_rds"] and got["n_ambig"] == want["n_ambig"]
        # every rank holds the multimapped reads of ITS shard (global read indices): in rank order they are the stream's
        parts = [None] * world
        dist.all_gather_object(parts, tuple(np.asarray(x) for x in got["multimapped"]))
        tax = np.concatenate([p[1] for p in parts])
        hl = np.concatenate([p[2] for p in parts])
        rd = np.concatenate([p[3] for p in parts])
        off, base = [np.zeros(1, dtype=np.uint64)], 0
        for p in parts:
            off.append(np.asarray(p[0][1:], dtype=np.uint64) + np.uint64(base))
            base += int(p[0][-1])
        off = np.concatenate(off)
        assert np.array_equal(tax, want["mm_tax"]) and np.array_equal(hl, want["mm_hitlen"]), (rank, idx)
        assert np.array_equal(rd, want["mm_read"]) and np.array_equal(off, want["mm_offsets"]), (rank, idx)
    if ("distinct_hint_ppm" in os.environ.get("MG_TEST_KNOBS", "")):
        assert getattr(job, "words_redone", 0) >= 1, getattr(job, "words_redone", 0)
# ---- the reference pipeline (reads sketched at the largest k only; prefix bitmaps OR-ed across the ranks), both hash modes ----
# ... and both ways a read k-mer meets a sketched one: by its hash (the table sharded by hash range), by what it is (every rank the
# whole table and its own reads; the ranks' counters all-gathered two bits a pair, or — counters that do not saturate — all-reduced)
REFPIPE_CASES = (([21, 31, 51], 0, "hash", 3), ([30, 40, 50, 60], 1, "hash", 3), ([31], 0, None, 3), ([21, 31, 51], 0, "kmer", 3),
                 ([30, 40, 50, 60], 1, "kmer", 0), ([31], 0, "kmer", 2))
if world > 3 or "distinct_hint_ppm" in os.environ.get("MG_TEST_KNOBS", ""):  # (the GPU tier's time: every case at world 2 and 3, the two headline ones elsewhere)
    REFPIPE_CASES = (REFPIPE_CASES[0], REFPIPE_CASES[3])
for ks, mode, match, cs in REFPIPE_CASES:
    hip.set_hash_mode(mode)
    hip.count_saturation(cs)  # (the columns below hold for every saturation value at or above ci = 2)
    oracle.set_hash_mode(mode)
    h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], 1000 if ("distinct_hint_ppm" in os.environ.get("MG_TEST_KNOBS", "")) else 200)
    table = hip.refdb_build(h, khi, klo, o, ks)
    full = table.download(kmers=match == "kmer")
    table.free()
    wtab = oracle.refpipe_build(*oracle.sketch_genomes_kmers(gb, go, ks[-1], 1000 if ("distinct_hint_ppm" in os.environ.get("MG_TEST_KNOBS", "")) else 200), ks)
    qh, qc, _, _ = oracle.sketch_reads(rb, ro, ks[-1], hmax=int(h.max()))
    whits, wsizes = oracle.refpipe_containment(qh, qc, 2, wtab)
    job = ShardJob(hip, dist, rank, world, k=ks, definition="reference_pipeline", match=match)
    job.load(my_rb, my_ro, my_recs, ref2tax, full)
    assert job.match == (match or "hash"), job.match  # (None: this table holds no k-mers -> the hash path)
    for idx, got in enumerate([job.step(), job.run(4), job.step()]):
        assert np.array_equal(got["hits_k"], whits) and np.array_equal(got["sizes_k"], wsizes), (rank, idx, ks, mode)
        for key in ("count", "bases", "first_seen"):
            assert np.array_equal(got[key], want[key]), (rank, idx, key)
        assert got["tot_rds"] == want["tot_rds"] and got["n_ambig"] == want["n_ambig"]
        assert got["definition"] == "reference_pipeline" and got["sketched_ks"] == [ks[-1]] and got["match"] == job.match
    assert whits[-1].max() > 100
    if match == "kmer":
        assert job.traffic_per_pass()["kmer_counts_all_gather"] > 0
    hip.set_hash_mode(0)
    hip.count_saturation(3)
    oracle.set_hash_mode(0)
dist.barrier()
dist.destroy_process_group()
print("two-ranks-one-gpu ok (rank %d)" % rank, flush=True)
