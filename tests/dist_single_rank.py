"""Helper of tests/test_pipeline_gpu.py: ONE rank, every collective of the multi-GPU exchange in the path (RCCL,
world size 1), pipelined ShardJob.run against the plain single-shard step and the oracle.  Run as a script."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")

import torch  # noqa: E402  (first: the library then binds to the same HIP runtime)
import torch.distributed as dist  # noqa: E402

import oracle  # noqa: E402
import util  # noqa: E402
from metalign_amd import synth  # noqa: E402
from metalign_amd._hip import Hip, debug_set  # noqa: E402
from metalign_amd.distributed import ShardJob  # noqa: E402

for _kv in filter(None, os.environ.get("MG_TEST_KNOBS", "").split(",")):  # (the parent test's knobs: the library reads no environment)
    debug_set(_kv.split("=")[0], int(_kv.split("=")[1]))

torch.cuda.set_device(0)
stream = torch.cuda.Stream()  # explicit: the default stream's handle is 0 and is refused by mg_init_on_stream
torch.cuda.set_stream(stream)
dist.init_process_group("nccl", rank=0, world_size=1)
hip = Hip.get(0, stream=stream.cuda_stream)
oracle.build()
gb, go = synth.make_genomes(60, 20000)
rb, ro, src = synth.make_reads(gb, go, 80000, npresent=9)
recs = synth.make_alignment_records(src + 1, 61)
ref2tax = np.arange(61, dtype=np.uint32)
n = 1000 if ("distinct_hint_ppm" in os.environ.get("MG_TEST_KNOBS", "")) else 200  # forced-overflow run: enough distinct hashes to fill a minimum-size table
want = oracle.profile_assign(recs, ref2tax, 61, 0.5)
# one k given bare (the single-k surface), then the k set of BASELINE configs[2] (the fused stage-A launch): every k's
# words in the one all-gather, its slices in the all-to-all round, its hits / sizes in the one all-reduce
for kspec in (21, [21, 31, 51]):
    ks = [kspec] if np.isscalar(kspec) else kspec
    tabs = [hip.sketch_genomes(gb, go, k, n) for k in ks]
    job = ShardJob(hip, dist, 0, 1, k=kspec, always_exchange=True)
    if np.isscalar(kspec):
        job.load(rb, ro, recs, ref2tax, tabs[0][0], tabs[0][1])
    else:
        job.load(rb, ro, recs, ref2tax, [t[0] for t in tabs], [t[1] for t in tabs])
    outs = [job.step(want_multimapped=True), job.run(4, want_multimapped=True), job.step(want_multimapped=True)]
    for idx, got in enumerate(outs):
        assert got["hits_k"].shape == (len(ks), 60)
        for ki, k in enumerate(ks):
            dbh, dbo = tabs[ki]
            oh, oc, otr, _ = oracle.sketch_reads(rb, ro, k, hmax=int(dbh.max()))
            nfiltered = util.job_sketch_size(oracle, job, ki, rb, ro, k, dbh)  # the job sketches through the table's filter or index
            ohits, osizes = oracle.containment(oh, oc, otr, 2, dbh, dbo)
            if not (np.array_equal(got["hits_k"][ki], ohits) and np.array_equal(got["sizes_k"][ki], osizes)):
                bad = np.nonzero(got["hits_k"][ki] != ohits)[0]
                print("MISMATCH in output", idx, "k", k, "hits differ at", len(bad), "genomes; sizes equal:",
                      np.array_equal(got["sizes_k"][ki], osizes), "sketch", got["sketch_sizes"][ki], nfiltered,
                      "sample", [(int(g), int(got["hits_k"][ki][g]), int(ohits[g])) for g in bad[:6]])
            assert np.array_equal(got["hits_k"][ki], ohits) and np.array_equal(got["sizes_k"][ki], osizes)
            assert got["sketch_sizes"][ki] == nfiltered, (k, got["sketch_sizes"][ki], nfiltered)
        assert np.array_equal(got["hits"], got["hits_k"][-1])
        for key in ("count", "bases", "first_seen"):
            assert np.array_equal(got[key], want[key]), key
        assert got["tot_rds"] == want["tot_rds"] and got["n_ambig"] == want["n_ambig"]
        off, tax, hl, rd = got["multimapped"]
        assert np.array_equal(off, want["mm_offsets"]) and np.array_equal(tax, want["mm_tax"])
        assert np.array_equal(hl, want["mm_hitlen"]) and np.array_equal(rd, want["mm_read"])
    if ("distinct_hint_ppm" in os.environ.get("MG_TEST_KNOBS", "")):  # the forced-overflow run must have taken the repeat-the-all-gather path
        assert getattr(job, "words_redone", 0) >= 4, getattr(job, "words_redone", 0)
dist.barrier()
dist.destroy_process_group()
print("dist-single-rank ok")
