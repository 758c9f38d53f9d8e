"""Helper of tests/test_gpu_multirank.py: bench.py's `--config 3` workload AT FULL SIZE as the N > 1 driver command runs it on a
rank — 12.5M reads, 15.6M alignment records, the 200k-genome table of 5 kb genomes (dense: ~22 % of all k-mers pass its
threshold, so the job builds the table's resident index and measures it against the bit filter at load) — with every
collective of the multi-GPU pass in the path (RCCL, world size 1: all-gather of the words, all-to-all of the sketch slices,
the reference pipeline's all-to-all of prefix-bitmap words, THE all-reduce), four passes in flight:

  * a >= 2M-read sample through ShardJob.step() against the C oracle on every host core: hits and sizes of ALL 200 000
    genomes for every k, every stage-C accumulator (bench.cpu_baseline_and_check with the job's collectives in the path);
  * the same with every list / counting table undersized (the knob distinct_hint_ppm): overflow -> redo at this size;
  * the full workload: run(3) == step(), sketch sizes and stage-C totals as they must be.

    python tests/dist_config3_full.py [reference_pipeline|sketch_per_k] [hash mode] [sample reads] [overflow|plain]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29537")

import torch  # noqa: E402  (first: the library then binds to the same HIP runtime)
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
import oracle  # noqa: E402
from metalign_amd import _hip  # noqa: E402
from metalign_amd._hip import Hip  # noqa: E402

definition = sys.argv[1] if len(sys.argv) > 1 else "reference_pipeline"
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
torch.cuda.set_device(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
dist.init_process_group("nccl", rank=0, world_size=1)
hip = Hip.get(0, stream=stream.cuda_stream)
oracle.build()
t0 = time.perf_counter()
match = sys.argv[5] if len(sys.argv) > 5 else None  # "kmer": stage A by k-mer identity (what bench.py --gpus N runs since round 6); None / "hash": by hash value
cfg = dict(bench.PRESETS[3], config=3, custom=False, definition=definition, hash_mode=mode, match=match)
w = bench.build_workload(cfg, 1000, 0, hip, definition, mode)
G, nreads = cfg["genomes"], len(w["ro"]) - 1
assert G == 200_000 and nreads == 12_500_000 and w["ntax"] == 10_001
report = {"definition": definition, "hash_mode": mode, "match": match, "build_s": time.perf_counter() - t0}
# ---- a sample against the oracle, every collective in the path; then once more with everything undersized ----
min_sample = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
args = argparse.Namespace(cpu_seconds=float(os.environ.get("MG_TEST_CPU_SECONDS", "6")), min_sample=min_sample)
for hint in ((None, "0.002") if (len(sys.argv) <= 4 or sys.argv[4] == "overflow") else (None,)):
    _hip.debug_set("distinct_hint_ppm", 0 if hint is None else int(float(hint) * 1e6))  # (lists, sketch buffers and counting tables too small: redo)
    t1 = time.perf_counter()
    base, check = bench.cpu_baseline_and_check(args, cfg, w, hip, dist=dist, force_dist=True)
    nsample = int(check["compared"].split("sample (")[1].split(" reads")[0])
    assert check["oracle_equal"], (hint, check["mismatch"])
    assert nsample >= min_sample, check["compared"]
    report["check_hint_%s" % hint] = {"sample_reads": nsample, "seconds": time.perf_counter() - t1, "cpu_reads_per_s": base["value"]}
_hip.debug_set("distinct_hint_ppm", 0)
# ---- the whole workload: the job the driver's N > 1 bench runs on a rank ----
job = bench.make_job(hip, dist, 0, 1, cfg, w, force_dist=True)
resident = sum(f.resident_bytes for f in job.engine.filters if f is not None)
report["resident_index_bytes"] = resident
report["resident_choice"] = getattr(job, "resident_choice", None)
one = job.step()
out = job.run(3)
assert np.array_equal(out["hits_k"], one["hits_k"]) and np.array_equal(out["sizes_k"], one["sizes_k"])
for key in ("count", "bases", "first_seen"):
    assert np.array_equal(out[key], one[key]), key
assert out["sketch_sizes"] == one["sketch_sizes"] and (out["tot_rds"], out["n_ambig"]) == (one["tot_rds"], one["n_ambig"])
assert job.match == (match or "hash") or definition != "reference_pipeline", job.match
assert out["hits_k"].shape == (3, G) and out["tot_rds"] == nreads
assert out["sketched_ks"] == ([51] if definition == "reference_pipeline" else [21, 31, 51])
# every genome that was sampled deeply is recovered at every k; the absent ones are not
depth = np.bincount(w["src"], minlength=G) * 150 / cfg["genome_len"]
ci = out["hits_k"] / np.maximum(out["sizes_k"], 1)
assert ci[-1][depth > 12].min() > 0.8 and ci[-1][depth == 0].max() < 0.05, (ci[-1][depth > 12].min(), ci[-1][depth == 0].max())
report["ms_per_pass_run3"] = None
hip.sync()
t2 = time.perf_counter()
job.run(5)
hip.sync()
report["ms_per_pass_run5"] = 1e3 * (time.perf_counter() - t2) / 5
report["total_s"] = time.perf_counter() - t0
dist.barrier()
dist.destroy_process_group()
print("config3-full ok " + json.dumps(report), flush=True)
