#!/usr/bin/env python3
"""bench.py — 150 bp reads/s through the hot path (CMash-style filter + profile) on N MI355X, inputs resident in HBM.

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
    stage A  the reads' k_max-mers counted against the table's BY K-MER IDENTITY (k_count_kmers: minimizer-partitioned, no hash
             on the read side — what kmc + kmc_tools intersect compute; --match hash: the read sketch of rounds 1-5,
             k_sketch_reads* + bucket sort / pack)                                         scripts/select_db.py:50-59
    stage B  containment, one column per k of the config                                  scripts/select_db.py:54-59,73-76
    stage C  assign + histogram (k_profile_pass), once                                    scripts/map_and_profile.py:193-264
Stage A/B definition (--definition):
    reference_pipeline (DEFAULT)  wired as the reference wires KMC and CMash: the reads are sketched at the LARGEST k only
                  (`kmc -k<kmax> -ci2 -cs3`, :50-52), stage B finds the matched sketched k_max-mers (:54-59) and derives every
                  smaller k's column from their k-prefixes (the streaming query, :73-76)
    sketch_per_k  rounds 1-3: every k has a genome sketch table of its own and the reads are hashed at every k (one fused launch)
--hash_mode 0 (default): MurmurHash3 of the lexicographically smaller strand, 64 bits; 1: min over the strands mod 9999999999971
(CMash as recollected).  The definitions NOT selected are measured in the same process over the same passes and reported beside
the headline (`definitions`).
File I/O, PCIe, on-device ingest and the host CAMI tail are NOT in the timed region; `with_ingest` reports the kept
command line end to end (files on disk -> subset DB / CAMI profile) on the SAME workload as a secondary figure.

Workloads = BASELINE.json configs:
  --config 1   1M reads vs 1k genomes (50 kb), k = 21                                  (the reference-sized case)
  --config 2   10M reads vs 10k genomes (50 kb), K = {21,31,51}, 12.5M alignment records, 10 001 taxa
               = the largest single-GPU configuration: the DEFAULT at N = 1
  --config 3   12.5M reads PER GPU vs a 200k-genome table (5 kb genomes: the dense regime), K = {21,31,51}, the table
               and the sample sketch sharded by hash range over the N ranks; at N = 8 this is configs[3]
               (100M reads): the DEFAULT at N > 1 (weak scaling in reads, the table is fixed)
  --config 4   config 3 + the CAMI profile of the reduced counts written once after the timed region (configs[4])
N > 1 is launched by torch.distributed.run, one rank per GPU (metalign_amd/distributed.py: reads and records sharded
contiguously, sketch tables AND sample sketches sharded by hash range; per pass one all-gather of a few words, one
all-to-all round of sketch slices, one all-reduce(sum, int64) of hits / sizes / per-taxon counts).

Prints ONE JSON line (rank 0).  `roofline` prices the dominant kernel family (k_sketch_reads*) against HBM with the
algorithmic bytes of SURVEY.md §8(d) — 158 B per read, ONE pass for all k — and, because that kernel is integer-VALU
bound, also gives `valu_frac` (VALU issue cycles / all SIMD cycles of the launch); `kernels` does the same for stage B
and stage C.  `cpu_baseline` is the C oracle (oracle/, a scalar port) on this host's cores over a bounded sample; the
same sample goes through the GPU path once more, untimed, and `check.oracle_equal` says whether every containment
count and every per-taxon accumulator came out identical.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_READ_K1 = 158  # 150 B of bases + 8 B offset, one pass for all k
ALGO_BYTES_PER_RECORD_K3 = 16
HBM_PEAK_GBS = 8000.0
SIMDS, XCDS = 1024, 8  # 256 CUs x 4 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs

PRESETS = {
    1: dict(reads=1_000_000, genomes=1000, genome_len=50_000, ks=[21], ntax=None,
            name="configs[1]: 1M synthetic 150bp reads vs 1k-genome sketch DB, k=21"),
    2: dict(reads=10_000_000, genomes=10_000, genome_len=50_000, ks=[21, 31, 51], ntax=None,
            name="configs[2]: 10M reads vs 10k-genome DB, multi-k {21,31,51} containment"),
    3: dict(reads=12_500_000, genomes=200_000, genome_len=5_000, ks=[21, 31, 51], ntax=10_001,
            name="configs[3]: 12.5M reads/GPU (100M at 8 GPUs) vs RefSeq-scale 200k-genome sketch DB"),
    4: dict(reads=12_500_000, genomes=200_000, genome_len=5_000, ks=[21, 31, 51], ntax=10_001,
            name="configs[4]: configs[3] + alignment replay -> full CAMI profile"),
}
# the reference's OWN parameters (scripts/select_db.py:44,50,69-70,75: kmc -k60, StreamingQueryDNADatabase 30-60-10, n = 1000) on
# configs[2]'s sizes: --preset stock makes it the headline; the default line carries it in `definitions`
STOCK_KS = [30, 40, 50, 60]
STOCK_NAME = "stock Metalign parameters (k_max = 60, k range 30-60-10, n = 1000) on configs[2]'s sizes"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--config", type=int, default=0, choices=[0, 1, 2, 3, 4], help="BASELINE.json config (0: 2 at N=1, 3 at N>1)")
    p.add_argument("--reads", type=int, default=0, help="reads per GPU (overrides the preset)")
    p.add_argument("--genomes", type=int, default=0)
    p.add_argument("--genome_len", type=int, default=0)
    p.add_argument("--ks", type=str, default="", help="comma-separated k-mer sizes (overrides the preset)")
    p.add_argument("--preset", choices=["", "stock"], default="", help="stock: the reference's own k_max = 60, range 30-60-10 on configs[2]'s sizes")
    p.add_argument("--sketch_n", type=int, default=1000)
    p.add_argument("--definition", choices=["reference_pipeline", "sketch_per_k"], default="reference_pipeline",
                   help="stage A/B as the reference wires it (reads sketched at the largest k only; default) or a sketch per k")
    p.add_argument("--hash_mode", type=int, choices=[0, 1], default=0, help="0: hash(min(kmer, revcomp)); 1: min(hash(kmer), hash(revcomp)) %% p")
    p.add_argument("--match", choices=["kmer", "hash"], default="kmer",
                   help="the reference pipeline on one GPU: a read k_max-mer meets a sketched one by what it IS (kmc's way, default) or by "
                        "its MurmurHash3 value (rounds 4-5)")
    p.add_argument("--no_definitions", action="store_true", help="skip the passes of the definitions that were not selected")
    p.add_argument("--no_cpu_baseline", action="store_true", help="also skips the oracle check (it shares the sample)")
    p.add_argument("--cpu_seconds", type=float, default=15.0, help="target CPU-baseline duration")
    p.add_argument("--no_kernel_table", action="store_true", help="skip the extra instrumented steps (clean traces)")
    p.add_argument("--no_secondary", action="store_true", help="skip the configs[1] secondary line and the with-ingest figure")
    p.add_argument("--no_same_workload_n1", action="store_true",
                   help="N > 1: do not measure this workload on one GPU first (rank 0 alone, same collectives in the path)")
    p.add_argument("--knob", action="append", default=[], metavar="KEY=VALUE",
                   help="A/B runs: a diagnostic knob of the library (mg_debug_set; include/metalign_hip.h), recorded in config.knobs")
    p.add_argument("--dry_run", action="store_true",
                   help="plumbing check only: rendezvous + the known-answer collectives, one JSON line, no GPU work")
    a = p.parse_args()
    return a


def resolve_config(args, world):
    cfg = args.config or (2 if max(world, args.gpus) == 1 else 3)
    pre = dict(PRESETS[cfg])
    if args.reads:
        pre["reads"] = args.reads
    if args.genomes:
        pre["genomes"] = args.genomes
    if args.genome_len:
        pre["genome_len"] = args.genome_len
    if args.ks:
        pre["ks"] = [int(x) for x in args.ks.split(",")]
    if args.preset == "stock":
        pre["ks"] = list(STOCK_KS)
        pre["name"] = STOCK_NAME
    pre["custom"] = bool(args.reads or args.genomes or args.genome_len or args.ks)
    pre["config"] = cfg
    pre["definition"], pre["hash_mode"] = args.definition, args.hash_mode
    pre["match"] = _match_for(args, pre["ks"]) if args.definition == "reference_pipeline" else None
    return pre


def _match_for(args, ks):
    """--match kmer (the default) means what the package does by default: identity from k_max = 25 on, the read sketch below."""
    from metalign_amd.distributed import kmer_match_by_default
    return "kmer" if (args.match == "kmer" and kmer_match_by_default(max(ks))) else "hash"


def build_tables(cfg, sketch_n, hip, gb, go, definition, hash_mode, world=1):
    """Stage A' on the GPU (not timed), under `hash_mode`: the genome sketch tables of `definition`."""
    prev = hip.hash_mode
    hip.set_hash_mode(hash_mode)
    try:
        if definition == "reference_pipeline":
            h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, cfg["ks"][-1], sketch_n)
            table = hip.refdb_build(h, khi, klo, o, cfg["ks"])
            arrays = table.download(kmers=cfg.get("match") == "kmer")  # (the k-mers in pair order: what the identity oracle looks up)
            arrays["max_hash"] = table.max_hash
            if world > 1 and cfg.get("match") != "kmer":  # (a rank uploads its share from the arrays; by k-mer identity every rank keeps the whole table)
                table.free()
                table = None
            return dict(definition=definition, hash_mode=hash_mode, ref_arrays=arrays, reftable=table, table_hashes=len(h))
        tables = [hip.sketch_genomes(gb, go, k, sketch_n) for k in cfg["ks"]]
        return dict(definition=definition, hash_mode=hash_mode, dbh=[t[0] for t in tables], dbo=[t[1] for t in tables],
                    table_hashes=sum(len(t[0]) for t in tables))
    finally:
        hip.set_hash_mode(prev)


def build_workload(cfg, sketch_n, rank, hip, definition="reference_pipeline", hash_mode=0, world=1):
    """Synthetic inputs, generated on the host once and left resident in HBM."""
    from metalign_amd import synth
    G = cfg["genomes"]
    gb, go = synth.make_genomes(G, cfg["genome_len"])  # the same table on every rank
    # 50 present genomes at configs[1] (1k genomes); one genome in 20 for the larger tables, so that the coverage per
    # present genome stays in a metagenome's range instead of growing into the thousands
    npresent = max(50, G // 20)
    rb, ro, src = synth.make_reads(gb, go, cfg["reads"], npresent=npresent, seed=synth.SEED + 1 + 1000 * rank)
    # accession rows: 0 = 'Unmapped', 1..G = one accession per genome
    recs = synth.make_alignment_records(src + 1, G + 1, seed=synth.SEED + 2 + 1000 * rank)
    if cfg["ntax"] is None:
        ref2tax = np.arange(G + 1, dtype=np.uint32)  # taxon row == accession row
    else:
        # the aligner runs against the SUBSET db that the pre-filter selected (10^2..10^4 taxa): dense ids
        ref2tax = (np.arange(G + 1, dtype=np.uint64) % np.uint64(cfg["ntax"])).astype(np.uint32)
    w = dict(rb=rb, ro=ro, src=src, recs=recs, ref2tax=ref2tax, ntax=int(ref2tax.max()) + 1, gb=gb, go=go)
    w.update(build_tables(cfg, sketch_n, hip, gb, go, definition, hash_mode, world))
    return w


def make_job(hip, dist, rank, world, cfg, w, force_dist=False, sub=None):
    """A job over workload `w` under w's definition and hash mode (the library's mode is set here and stays: a job's passes
    hash the reads by the definition its tables were sketched with)."""
    from metalign_amd import distributed as mgd
    hip.set_hash_mode(w["hash_mode"])
    exchange = dist is not None and (world > 1 or force_dist)
    match = None
    if w["definition"] == "reference_pipeline":
        match = "kmer" if (cfg.get("match") == "kmer" and 15 <= cfg["ks"][-1] <= 64) else "hash"
    job = mgd.ShardJob(hip, dist, rank, world, k=cfg["ks"], ci=2, pct_id=0.5, always_exchange=force_dist, definition=w["definition"],
                       match=match)
    rb, ro, recs = (w["rb"], w["ro"], w["recs"]) if sub is None else sub
    if w["definition"] == "reference_pipeline":
        job.load(rb, ro, recs, w["ref2tax"], w["ref_arrays"], ntax=w["ntax"], reftable=w["reftable"])
    else:
        job.load(rb, ro, recs, w["ref2tax"], w["dbh"], w["dbo"], ntax=w["ntax"])
    return job


def cpu_baseline_and_check(args, cfg, w, hip, dist=None, force_dist=False):
    """The CPU oracle on a bounded sample of the same workload (the first n reads and their alignment records) on every
    host core: the sample is cut into one contiguous share per thread (ctypes releases the GIL inside the C oracle;
    threads, not processes: this process has initialised the GPU); per share the read sketch of every k and stage C
    (the additive part: a timing does not need the edge fix-up of the carried state); then the shares' sketches are
    merged per k and run against the full tables.  The single-core rate of the same code is reported beside it.
    The SAME sample then goes through the GPU path once (untimed) and everything is compared: hits and sizes of every
    genome for every k, count / bases / first_seen of every taxon, tot_rds, n_ambig."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    oracle.build()
    oracle.set_hash_mode(w["hash_mode"])
    ks = cfg["ks"]
    refpipe = w["definition"] == "reference_pipeline"
    # the k the READS are sketched at: the reference pipeline counts only k_max-mers (kmc -k<kmax>, scripts/select_db.py:50-52)
    sk_ks = [ks[-1]] if refpipe else ks
    hmaxs = [int(w["ref_arrays"]["max_hash"])] if refpipe else [int(h.max()) for h in w["dbh"]]
    leaders = np.cumsum(w["recs"]["ref_new"] >> 31)
    nreads_all = len(w["ro"]) - 1
    # stage A by k-mer identity (the default): the oracle counts the reads' canonical k_max-mers among the table's k-mers as kmc +
    # kmc_tools intersect do (mgo_kmer_table_*: a look-up per window, no hash of the k-mer's text)
    by_kmer = refpipe and cfg.get("match") == "kmer" and dist is None and w["ref_arrays"].get("kmer_hi") is not None
    ktab = oracle.KmerTable(w["ref_arrays"]["kmer_hi"], w["ref_arrays"]["kmer_lo"], ks[-1]) if by_kmer else None

    def share(lo, hi):  # reads [lo, hi) and their records
        b0, b1 = int(w["ro"][lo]), int(w["ro"][hi])
        r0 = int(np.searchsorted(leaders, lo, side="right"))
        r1 = int(np.searchsorted(leaders, hi, side="right"))
        if by_kmer:
            sk = ktab.count(w["rb"][b0:b1], w["ro"][lo: hi + 1] - w["ro"][lo])[0]
        else:
            sk = [oracle.sketch_reads(w["rb"][b0:b1], w["ro"][lo: hi + 1] - w["ro"][lo], k, hmax=hm)[:2] for k, hm in zip(sk_ks, hmaxs)]
        prof = oracle.profile_assign(w["recs"][r0:r1], w["ref2tax"], w["ntax"], 0.5) if r1 > r0 else None
        return sk, prof

    def run(nreads, cores):
        t0 = time.perf_counter()
        cuts = [nreads * i // cores for i in range(cores + 1)]
        if cores == 1:
            parts = [share(0, nreads)]
        else:
            with ThreadPoolExecutor(cores) as ex:
                parts = list(ex.map(lambda i: share(cuts[i], cuts[i + 1]), range(cores)))
        merged = []
        if by_kmer:
            total = parts[0][0]
            for p in parts[1:]:
                total = total + p[0]
            hits, sizes = oracle.refpipe_containment_counts(ktab.per_pair(total, oracle.DEFAULT_CS), 2, w["ref_arrays"])
            return time.perf_counter() - t0, [(hits[ki], sizes[ki]) for ki in range(len(ks))]
        for ki in range(len(sk_ks)):
            allh = np.concatenate([p[0][ki][0] for p in parts])
            allc = np.concatenate([p[0][ki][1] for p in parts]).astype(np.uint64)
            uh, inv = np.unique(allh, return_inverse=True)
            uc = np.minimum(np.bincount(inv, weights=allc, minlength=len(uh)), oracle.DEFAULT_CS).astype(np.uint32)
            merged.append((uh, uc))
        if refpipe:
            # the oracle's query against the table the GPU builder laid out (the builder itself is held to the oracle's by
            # tests/test_gpu_refpipe.py and, at this size, tests/test_gpu_fullsize.py: 54 s of qsort are not for a benchmark)
            hits, sizes = oracle.refpipe_containment(merged[0][0], merged[0][1], 2, w["ref_arrays"])
            res = [(hits[ki], sizes[ki]) for ki in range(len(ks))]
        else:
            res = [oracle.containment(uh, uc, False, 2, w["dbh"][ki], w["dbo"][ki]) for ki, (uh, uc) in enumerate(merged)]
        return time.perf_counter() - t0, res

    cores = max(1, min(os.cpu_count() or 1, 64))
    probe = min(20000, nreads_all)
    t1, _ = run(probe, 1)
    single = probe / t1
    # (min_sample: the GPU test tier asks for a sample of its own size whatever the probe's rate — against a 200k-genome
    # table the probe's 20 000 reads mostly time the containment of 2 x 10^8 table hashes)
    n = int(min(nreads_all, max(probe, single * cores * 0.7 * args.cpu_seconds, getattr(args, "min_sample", 0))))
    t, want_hs = run(n, cores)
    base = {"value": n / t, "unit": "reads/s", "cores": cores, "kind": "port", "single_core_value": single,
            "sample": "the first %d of the %d reads (%.1f %%) + their alignment records in %d contiguous shares, one thread "
                      "each: %s for k in %s, merged, containment (%s) against the full %d-genome tables for k in %s, stage C; "
                      "C oracle, %.1f s" % (n, nreads_all, 100.0 * n / nreads_all, cores,
                                            "the reads' canonical k-mers counted among the table's k-mers by identity" if by_kmer else "read sketches",
                                            sk_ks, w["definition"], cfg["genomes"], ks, t)}
    # ---- the same sample through the GPU path (one job, one step), against the oracle ----
    b1 = int(w["ro"][n])
    r1 = int(np.searchsorted(leaders, n, side="right"))
    sub = (w["rb"][:b1], w["ro"][: n + 1], w["recs"][:r1])
    want_c = oracle.profile_assign(sub[2], w["ref2tax"], w["ntax"], 0.5)  # exact (sequential) stage C of the sample
    # (dist / force_dist: the same job with every collective of the multi-GPU pass in its path, at world size 1 —
    # tests/dist_config3_full.py)
    job = make_job(hip, dist, 0, 1, cfg, w, force_dist=force_dist, sub=sub)
    got = job.step()
    bad = []
    for ki, k in enumerate(ks):
        if not np.array_equal(got["hits_k"][ki], want_hs[ki][0]):
            bad.append("hits k=%d" % k)
        if not np.array_equal(got["sizes_k"][ki], want_hs[ki][1]):
            bad.append("sizes k=%d" % k)
    for key in ("count", "bases", "first_seen"):
        if not np.array_equal(got[key], want_c[key]):
            bad.append(key)
    if (got["tot_rds"], got["n_ambig"]) != (want_c["tot_rds"], want_c["n_ambig"]):
        bad.append("tot_rds/n_ambig")
    oracle.set_hash_mode(0)
    check = {"oracle_equal": not bad, "mismatch": bad, "definition": w["definition"], "hash_mode": w["hash_mode"], "match": got.get("match"),
             "compared": "hits and sizes of all %d genomes for k in %s; count / bases / first_seen of all %d taxa; tot_rds; "
                         "n_ambig — GPU path vs C oracle on the cpu_baseline sample (%d reads, %d records)"
                         % (cfg["genomes"], ks, w["ntax"], n, r1)}
    return base, check


def committed_profile(name, cfg):
    """Per-launch counters of the committed rocprofv3 PMC passes (a benchmark cannot profile itself):
    profiles/<round>/<name>, newest round whose workload matches this run; else None."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for rnd in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        for fn in sorted(os.listdir(os.path.join(pdir, rnd))):  # <name> itself or <workload tag>_<name>
            if not fn.endswith(name):
                continue
            with open(os.path.join(pdir, rnd, fn)) as fh:
                d = json.load(fh)
            wl = d.get("workload", {})
            # (profiles of rounds 1-3 carry no definition: a sketch per k, hash mode 0)
            # (... and those of rounds 4-5 no match: the reference pipeline met k-mers by hash value then)
            refp = wl.get("definition", "sketch_per_k") == "reference_pipeline"
            if (wl.get("reads"), wl.get("genomes"), wl.get("ks", [wl.get("k")]), wl.get("definition", "sketch_per_k"), wl.get("hash_mode", 0),
                    (wl.get("match") or "hash") if refp else None) == \
                    (cfg["reads"], cfg["genomes"], cfg["ks"], cfg.get("definition", "sketch_per_k"), cfg.get("hash_mode", 0),
                     (cfg.get("match") or "hash") if cfg.get("definition") == "reference_pipeline" else None):
                best = d
    return best


def committed_kernel_stats(cfg):
    """The dominant kernel's average duration in the committed `rocprofv3 --kernel-trace --stats` summary of this workload
    (profiles/<round>/config<N>_kernel_stats.csv, newest round): the cross-check of roofline.kernel_ms_alone."""
    import csv
    pdir = os.path.join(ROOT, "profiles")
    best = None
    kmax = cfg["ks"][-1] if cfg.get("definition") == "reference_pipeline" else None
    for rnd in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        fn = os.path.join(pdir, rnd, "config%d_kernel_stats.csv" % cfg["config"])
        if not os.path.isfile(fn):
            continue
        with open(fn) as fh:
            for row in csv.DictReader(fh):
                name = row.get("kernel", row.get("Name", ""))
                fam = "k_count_kmers" if cfg.get("match") == "kmer" and kmax is not None else "k_sketch_reads"
                if fam in name and (kmax is None or ("<%d," % kmax) in name or ("<%d>" % kmax) in name):
                    best = {"kernel": name, "avg_ms": float(row.get("avg_ns", row.get("AverageNs", 0))) / 1e6,
                            "calls": int(row.get("calls", row.get("Calls", 0))),
                            "source": "profiles/%s/config%d_kernel_stats.csv" % (rnd, cfg["config"])}
                    break
    return best


def valu_roofline(sq, ms_alone_live):
    """valu_frac and what it was computed from; {} of Nones when a piece is missing."""
    out = {"valu_frac": None, "valu_model": None}
    if not sq:
        return out
    sa = sq.get("stage_a", sq.get("k_sketch_reads"))
    names = sa["kernels"]
    insts = sa.get("SQ_INSTS_VALU_per_pass")
    gui = sum(sq["kernels"][k].get("GRBM_GUI_ACTIVE", 0.0) for k in names)
    model = None
    pdir = os.path.join(ROOT, "profiles")
    # the per-opcode table of the kernel the passes run: the fused {21,31,51} launch (k1_valu_roofline.json) or the one-k kernel
    # of the reference pipeline's largest k (k1_single_k<K>_valu_roofline.json)
    fused = len(names) == 1 and "multi" in names[0]
    m = re.match(r"k_sketch_reads<(\d+)", names[0]) if len(names) == 1 else None
    model_name = "k1_valu_roofline.json" if fused else ("k1_single_k%s_valu_roofline.json" % m.group(1) if m else None)
    mk = re.match(r"k_count_kmers<(\d+)", names[0]) if len(names) == 1 else None
    if mk:
        model_name = "k1_count_kmers_k%s_valu_roofline.json" % mk.group(1)
    for rnd in sorted(os.listdir(pdir)) if (os.path.isdir(pdir) and model_name) else []:
        fn = os.path.join(pdir, rnd, model_name)
        if os.path.isfile(fn):
            with open(fn) as fh:
                model = dict(json.load(fh), source="profiles/%s/%s" % (rnd, model_name))
    if not (insts and gui and model):
        return out
    simd_cycles = gui / XCDS * SIMDS
    cpi = model["cycles_per_valu_instruction"]
    out["valu_frac"] = insts * cpi / simd_cycles
    out["valu_model"] = {"cycles_per_valu_instruction": cpi, "valu_instructions_per_wave_step_static": model["valu_instructions_per_wave_step"],
                         "source": model["source"], "SQ_INSTS_VALU": insts, "GRBM_GUI_ACTIVE": gui,
                         "simd_cycles_of_the_launch": simd_cycles,
                         "engine_clock_GHz_in_that_pass": (gui / XCDS) / (sq["kernels"][names[0]].get("duration_ns", 0.0) or float("nan")) if sq["kernels"][names[0]].get("duration_ns") else None,
                         "ms_per_pass_alone_live": ms_alone_live}
    return out


def committed_run(name, cfg):
    """value / ms_per_step of a committed bench line (profiles/<round>/<name>, newest round) if it ran this per-GPU
    workload; else None."""
    pdir = os.path.join(ROOT, "profiles")
    best = None
    for rnd in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        fn = os.path.join(pdir, rnd, name)
        if not os.path.isfile(fn):
            continue
        with open(fn) as fh:
            d = json.loads(fh.read().strip().splitlines()[-1])
        wl = d.get("config", {}).get("workload", "")
        if ("%d synthetic 150bp reads/GPU vs %d-genome" % (cfg["reads"], cfg["genomes"])) in wl and ("k in %s" % cfg["ks"]) in wl:
            best = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "n_gpus": d["n_gpus"],
                    "source": "profiles/%s/%s" % (rnd, name),
                    "note": "bench.py --config %d at world size 1 with the exchange path forced (MG_FORCE_DIST=1): the "
                            "denominator for weak-scaling efficiency of this workload; bench.py's DEFAULT at N = 1 is "
                            "configs[2], a different workload" % cfg["config"]}
    return best


def secondary_config1(hip, args):
    """configs[1] (1M reads, 1k genomes, k = 21) in the same process: reads/s of the pipelined passes."""
    cfg = dict(PRESETS[1], config=1, match=_match_for(args, PRESETS[1]["ks"]) if args.definition == "reference_pipeline" else None)
    w = build_workload(cfg, args.sketch_n, 0, hip, args.definition, args.hash_mode)
    job = make_job(hip, None, 0, 1, cfg, w)
    job.run(60)
    hip.sync()
    steps = 200
    t0 = time.perf_counter()
    job.run(steps)
    hip.sync()
    dt = time.perf_counter() - t0
    if w.get("reftable") is not None:
        del job
        w["reftable"].free()
    hip.set_hash_mode(0)
    return {"workload": cfg["name"], "value": cfg["reads"] / (dt / steps), "unit": "reads/s", "ms_per_step": 1e3 * dt / steps,
            "steps": steps}


def other_definitions(hip, args, cfg, w):
    """The stage A/B definitions that were NOT selected, on the same reads / records / genomes, over the same number of
    pipelined passes (untimed by the driver; the same loop as the headline's): ms per pass and reads/s each."""
    out = {}
    todo = [("reference_pipeline", 0, None, "kmer"), ("reference_pipeline", 0, None, "hash"), ("reference_pipeline", 1, None, "kmer"),
            ("reference_pipeline", 1, None, "hash"), ("sketch_per_k", 0, None, None)]
    if cfg["ks"] != STOCK_KS:  # the reference's own k set, both hash definitions (the only configuration a Metalign user runs)
        todo += [("reference_pipeline", 0, STOCK_KS, "kmer"), ("reference_pipeline", 0, STOCK_KS, "hash"), ("reference_pipeline", 1, STOCK_KS, "kmer")]
    for definition, mode, ks, match in todo:
        if (definition, mode, match) == (args.definition, args.hash_mode, cfg.get("match")) and ks is None:
            continue
        t0 = time.perf_counter()
        w2 = dict(w)
        for key in ("ref_arrays", "reftable", "dbh", "dbo"):
            w2.pop(key, None)
        cfg_run = dict(cfg, match=match) if ks is None else dict(cfg, ks=list(ks), name=STOCK_NAME, match=match)
        k1_name = "count_kmers" if match == "kmer" else "sketch_reads"
        w2.update(build_tables(cfg_run, args.sketch_n, hip, w["gb"], w["go"], definition, mode))
        job = make_job(hip, None, 0, 1, cfg_run, w2)
        job.run(max(args.warmup, 2))
        hip.sync()
        hip.prof_reset()
        hip.prof_enable(True)
        hip.prof_only(k1_name)
        t1 = time.perf_counter()
        res = job.run(args.steps)
        hip.sync()
        dt = (time.perf_counter() - t1) / args.steps
        nk1, k1_ms = hip.prof_get(k1_name)
        hip.prof_enable(False)
        key = ("stock_" if ks is not None else "") + "%s_mode%d" % (definition, mode) + ("_hash_match" if match == "hash" else "")
        out[key] = {
            "definition": definition, "hash_mode": mode, "match": res.get("match"), "ks": cfg_run["ks"], "ms_per_pass": 1e3 * dt, "value": cfg["reads"] / dt, "unit": "reads/s",
            "steps": args.steps, "stage_a_avg_launch_ms": k1_ms / max(nk1, 1), "stage_a_launches_per_pass": nk1 / max(args.steps, 1),
            "sketched_ks": res.get("sketched_ks"), "sketch_sizes": res.get("sketch_sizes"), "top_genomes_recovered": res.get("top_ok"),
            "setup_s": t1 - t0}
        del job
        if ks is not None and mode == 0 and match == "kmer" and not args.no_cpu_baseline:
            # the stock preset against the oracle on a sample of its own (all 10 000 genomes x {30,40,50,60}, every stage-C accumulator)
            import copy
            a2 = copy.copy(args)
            a2.cpu_seconds = min(args.cpu_seconds, 6.0)
            base2, chk2 = cpu_baseline_and_check(a2, cfg_run, w2, hip)
            out[key]["check"] = chk2
            out[key]["cpu_baseline"] = {k2: base2[k2] for k2 in ("value", "unit", "cores", "kind")}
        if w2.get("reftable") is not None:
            w2["reftable"].free()
        hip.mem_trim()
    hip.set_hash_mode(w["hash_mode"])
    return out


_LAUNCH_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK",
               "ROLE_WORLD_SIZE", "ROLE_NAME", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RESTART_COUNT",
               "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING",
               "TORCHELASTIC_ERROR_FILE")


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _relay(cmd, env, timeout=None):
    """Runs a child bench, passes its stderr through, returns (rc, the LAST JSON line it printed or None).  The caller
    has not touched the GPU (a process that has must not be the parent of a launcher on this pool)."""
    import subprocess
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    try:
        out, _ = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        out, _ = proc.communicate()
    for ln in (out or "").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    return proc.returncode, line


def self_launch(args):
    """`python3 bench.py --gpus N` as a plain command (no WORLD_SIZE in the environment): start the N ranks ourselves —
    python -m torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1 — BEFORE anything in this process
    touches the GPU, and relay rank 0's one JSON line.  Under torchrun (the driver's N > 1 command) this is not taken."""
    env = {k: v for k, v in os.environ.items() if k not in _LAUNCH_ENV}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    rc, line = _relay(cmd, env)
    if line:
        print(line, flush=True)
    return rc if rc else (0 if line else 1)


def same_workload_on_one_gpu(args, cfg):
    """The per-GPU workload of this N > 1 run on ONE GPU of this node, in this run: a child process (rank 0's GPU, world
    size 1, MG_FORCE_DIST=1: every collective of the N-rank pass stays in the path, the table is whole), started by rank 0
    BEFORE it initialises its GPU while the other ranks wait in the rendezvous.  Weak-scaling efficiency at N is
    value(N) / (N x this value); bench.py's own N = 1 default is configs[2], a different workload."""
    env = {k: v for k, v in os.environ.items() if k not in _LAUNCH_ENV}
    env.update(MG_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", str(cfg["config"]), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--sketch_n", str(args.sketch_n), "--no_cpu_baseline", "--no_secondary", "--no_kernel_table"]
    for flag, val in (("--reads", args.reads), ("--genomes", args.genomes), ("--genome_len", args.genome_len), ("--ks", args.ks)):
        if val:
            cmd += [flag, str(val)]
    t0 = time.perf_counter()
    rc, line = _relay(cmd, env, timeout=900)
    if rc or not line:
        return {"error": "the one-GPU run of this workload failed (rc %r)" % rc}
    d = json.loads(line)
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "n_gpus": 1, "steps": d["steps"],
            "source": "this run: rank 0's GPU alone before the group run, world size 1 with the exchange path forced "
                      "(MG_FORCE_DIST=1), %.0f s wall" % (time.perf_counter() - t0),
            "workload": d["config"]["workload"]}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    force_dist = os.environ.get("MG_FORCE_DIST") == "1"  # single-GPU validation of the torch/RCCL path
    if args.gpus > 1 and world != args.gpus:
        print("bench.py: --gpus %d but the launcher started %d ranks: reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)
    n1 = None
    if world > 1 and rank == 0 and not args.no_same_workload_n1 and not args.dry_run:
        n1 = same_workload_on_one_gpu(args, resolve_config(args, world))  # before this process touches the GPU
    if args.dry_run:
        import datetime
        import torch
        import torch.distributed as dist
        from metalign_amd import distributed as mgd
        backend = os.environ.get("MG_DIST_BACKEND", "nccl")
        if world > 1 or force_dist:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
            dev = "cpu"
            if backend == "nccl":
                torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
                dev = "cuda"
            mgd.selfcheck_collectives(dist, torch, rank, world, dev)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "150bp reads/s end-to-end (CMash filter + profile)", "value": None, "unit": "reads/s",
                              "n_gpus": world, "steps": 0, "warmup": 0, "dry_run": True,
                              "collective_selfcheck": "ok" if (world > 1 or force_dist) else "skipped (one rank)",
                              "backend": backend}), flush=True)
        return
    if world > 1 or args.gpus > 1 or force_dist:
        # torch first: the library then binds to the same HIP runtime and launches on torch's stream
        import datetime
        import torch
        import torch.distributed as dist
        local_rank %= max(torch.cuda.device_count(), 1)  # (MG_DIST_BACKEND=gloo: several ranks on one GPU, for validation)
        torch.cuda.set_device(local_rank)
        # an explicit stream shared by torch (collectives synchronise with it) and the library's main stream
        torch_stream = torch.cuda.Stream()
        torch.cuda.set_stream(torch_stream)
        # RCCL prints its version banner on stdout when the first communicator comes up: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            # (the timeout covers rank 0's one-GPU run of the workload, during which the other ranks wait here)
            dist.init_process_group(os.environ.get("MG_DIST_BACKEND", "nccl"), rank=rank, world_size=world,
                                    timeout=datetime.timedelta(minutes=30))
            dist.barrier()
            torch.cuda.synchronize()
            if world > 1:  # known-answer collectives (uneven all-to-all, int64 all-reduce): fail loudly at start-up
                from metalign_amd import distributed as mgd
                mgd.selfcheck_collectives(dist, torch, rank, world, "cuda")
        finally:
            sys.stdout.flush()
            import ctypes
            ctypes.CDLL(None).fflush(None)  # the banner sits in C stdio's buffer
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
        from metalign_amd._hip import Hip
        hip = Hip.get(local_rank, stream=torch_stream.cuda_stream)
    else:
        from metalign_amd._hip import Hip
        hip = Hip.get(0)

    knobs = {}
    for kv in args.knob:
        key, _, val = kv.partition("=")
        from metalign_amd import _hip as _hipmod
        _hipmod.debug_set(key, int(val or 1))
        knobs[key] = int(val or 1)
    cfg = resolve_config(args, world)
    w = build_workload(cfg, args.sketch_n, rank, hip, args.definition, args.hash_mode, world)
    job = make_job(hip, dist, rank, world, cfg, w, force_dist)
    nreads, nrecs, K = cfg["reads"], len(w["recs"]), len(cfg["ks"])

    def sync():
        hip.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    # The GPU has idled through the workload generation above and takes tens of milliseconds of work to come back to
    # full clocks: untimed ramp-up passes first (about 0.1 s of GPU time), so that the figure does not depend on how
    # small W is; then the W warm-up passes.
    job.run(max(2, int(2e6 * 60 / max(nreads * K, 1))))
    sync()
    job.run(args.warmup)
    sync()
    hip.prof_reset()
    hip.prof_enable(True)
    k1_name = "count_kmers" if getattr(job, "match", None) == "kmer" else "sketch_reads"
    hip.prof_only(k1_name)  # the dominant kernel is timed with HIP events inside the timed region
    t0 = time.perf_counter()
    out = job.run(args.steps)  # K passes, software-pipelined (stage A of pass i+1 is queued before pass i is finished)
    sync()
    dt = time.perf_counter() - t0
    nk1, k1_ms = hip.prof_get(k1_name)
    # per-kernel table from a few extra, untimed steps on ONE stream (nothing overlaps: clean per-family times)
    kernels_ms = {}
    if not args.no_kernel_table:
        hip.prof_reset()
        hip.prof_enable(True)
        hip.stage_c_side_stream(False)
        nt = min(args.steps, 3)
        for _ in range(nt):
            job.step()
        sync()
        hip.stage_c_side_stream(True)
        for name in ("table_clear", "count_kmers", "sketch_reads", "merge_insert", "merge_sort", "bucket_sort", "bucket_pack", "sketch_sort",
                     "sketch_rle", "contain_index", "containment", "refpipe_count", "profile_map", "profile_pass"):
            n, t = hip.prof_get(name)
            if n:
                kernels_ms[name] = {"ms_per_pass": round(t / nt, 4), "launches_per_pass": n / nt}
    hip.prof_enable(False)
    if dist is not None:
        import torch
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms = 1e3 * dt / args.steps
        k1_per_pass_ms = k1_ms / max(args.steps, 1)  # all k of one pass (one fused launch, or one launch per k)
        launches_per_pass = nk1 / max(args.steps, 1)
        algo_k1 = ALGO_BYTES_PER_READ_K1 * nreads
        # the kernel's OWN duration: HIP events around it with nothing else on the device (the extra one-stream steps below the
        # timed region; rocprofv3's average of the same kernel is in profiles/<round>/config2_kernel_stats.csv).  In the timed
        # region two launches overlap (stage A of pass i + 1 starts while pass i finishes), so a launch there is STRETCHED
        # (avg_launch_ms_pipelined) although one starts every period_ms: that is why ms_per_step can be below the kernel alone.
        k1_alone_ms = kernels_ms.get(k1_name, {}).get("ms_per_pass") or k1_per_pass_ms
        achieved = algo_k1 / (k1_alone_ms * 1e-3) / 1e9 if k1_alone_ms else 0.0
        traffic = committed_profile("pmc_traffic.json", cfg)
        sq = committed_profile("pmc_sq_summary.json", cfg)
        by_kmer = k1_name == "count_kmers"
        sq_a = (sq.get("stage_a", sq.get("k_sketch_reads")) if sq else None)
        tr_a = (traffic.get("stage_a", traffic.get("k_sketch_reads")) if traffic else None)
        valu_insts = sq_a.get("SQ_INSTS_VALU_per_pass") if sq_a else None
        roof = {"kernel": ("k_count_kmers<%d> (stage A by k-mer identity: %.0f launch(es) per pass)" % (cfg["ks"][-1], launches_per_pass)) if by_kmer
                else "k_sketch_reads* (stage A, all k of the pass: %.0f launch(es) per pass)" % launches_per_pass,
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": (tr_a.get("hbm_bytes_per_pass", tr_a.get("hbm_bytes_per_launch")) if tr_a else None),
                "kernel_ms_alone": k1_alone_ms, "period_ms": ms,
                "avg_launch_ms_pipelined": k1_ms / max(nk1, 1), "overlap_ms": max(0.0, k1_per_pass_ms - ms),
                "ms_per_pass_pipelined": k1_per_pass_ms,
                "rocprofv3": committed_kernel_stats(cfg),
                "algorithmic_bytes_per_pass": algo_k1,
                "traffic_correction": (traffic or {}).get("correction"),
                # VALU roofline (the kernel is integer-VALU bound), calibrated: every opcode of the hot loop priced with its
                # measured issue cost (tools/ubench_valu.hip -> profiles/<round>/valu_classes.json; tools/valu_roofline.py
                # -> k1_valu_roofline.json: cycles per VALU instruction of this kernel's mix), times the VALU instructions the
                # launch executed, over the SIMD cycles the launch had — both from the SAME committed PMC pass
                # (SQ_INSTS_VALU; GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), the kernel running alone on one stream
                **valu_roofline(sq, kernels_ms.get(k1_name, {}).get("ms_per_pass")),
                "ms_per_pass_alone": kernels_ms.get(k1_name, {}).get("ms_per_pass"),
                "valu_insts_per_pass": valu_insts,
                "note": (("stage A by k-mer identity (k_count_kmers): one lane per read slides a 15-mer minimizer over its windows — no k-mer is "
                          "hashed — and leaves one event per run of windows; per run the 19 bases around the minimizer are hashed and ONE bit "
                          "of a 2^29-bit gate is probed (6.9 runs per 150 bp read: 6.9 x 10^7 random 64-byte sectors per 10M reads), the 4 % "
                          "that pass read their bucket's 128-byte line. achieved / frac = 158 B/read x reads / kernel_ms_alone against 8 TB/s; "
                          "traffic (FETCH_SIZE + WRITE_SIZE of the committed PMC passes) is 4.5 x that: the gate's sectors. The walk alone "
                          "is 1.3 of the kernel's 2.1-2.2 ms (SQ_INSTS_VALU 33 per wave-step, three wavefronts per SIMD at 168 VGPRs); "
                          "valu_frac = SQ_INSTS_VALU x 4 cycles (every opcode of the walk's step is full rate: valu_model.source says how that was "
                          "taken) / the SIMD cycles of the launch (GRBM_GUI_ACTIVE / 8 x 1024), from the committed PMC pass: the kernel is an "
                          "integer-VALU kernel with a latency tail (profiles/r06/kcount_*_final.txt hold its SQ counters). ") if by_kmer else
                         ("integer-VALU bound (one MurmurHash3 per k-mer per k): valu_frac = SQ_INSTS_VALU x the measured issue "
                          "cycles per VALU instruction of this kernel's opcode mix / the SIMD cycles of the launch (GRBM_GUI_ACTIVE "
                          "/ 8 x 1024), all from the committed PMC pass of this workload, is the figure that describes it; achieved "
                          "/ frac = 158 B/read x reads / kernel_ms_alone (the kernel's own duration, HIP events, nothing else on the "
                          "device) against 8 TB/s; ")) +
                        "in the timed region a launch starts every period_ms and overlaps its predecessor by overlap_ms "
                        "(avg_launch_ms_pipelined is the stretched duration there)"}
        kern = []
        if "containment" in kernels_ms:
            t_b = kernels_ms["containment"]["ms_per_pass"] + kernels_ms.get("contain_index", {}).get("ms_per_pass", 0.0)
            t_b += kernels_ms.get("refpipe_count", {}).get("ms_per_pass", 0.0)
            # 8 B per table hash and k; the reference pipeline: the largest k's hashes, and per smaller k one
            # 8-byte (prefix number, genome) entry per table hash
            algo = (w["table_hashes"] * (len(cfg["ks"]) if w["definition"] == "reference_pipeline" else 1)) * 8 // max(world, 1)
            kern.append({"kernel": "k_contain_pairs (+ index, reduce%s), all k" % (", k_refpipe_count" if w["definition"] == "reference_pipeline" else ""),
                         "algorithmic_bytes_per_pass": algo, "ms_per_pass": t_b,
                         "achieved_GBs": algo / (t_b * 1e-3) / 1e9, "frac": algo / (t_b * 1e-3) / 1e9 / HBM_PEAK_GBS, "bound": "hbm"})
        if "profile_pass" in kernels_ms:
            t_c = kernels_ms["profile_pass"]["ms_per_pass"]
            algo = ALGO_BYTES_PER_RECORD_K3 * nrecs
            kern.append({"kernel": "k_profile_pass", "algorithmic_bytes_per_pass": algo, "ms_per_pass": t_c,
                         "achieved_GBs": algo / (t_c * 1e-3) / 1e9, "frac": algo / (t_c * 1e-3) / 1e9 / HBM_PEAK_GBS, "bound": "hbm"})
        res = {
            "metric": "150bp reads/s end-to-end (CMash filter + profile)",
            "value": nreads * world / (dt / args.steps),
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "%s%s — %d synthetic 150bp reads/GPU vs %d-genome sketch DB (%d bp genomes, n=%d), "
                                   "k in %s, %d alignment records/GPU, %d taxa, 1 MI355X per rank; inputs resident in HBM"
                                   % (cfg["name"], " [custom sizes]" if cfg["custom"] else "", nreads, cfg["genomes"],
                                      cfg["genome_len"], args.sketch_n, cfg["ks"], nrecs, w["ntax"]),
                       "baseline_config": cfg["config"],
                       "stage_a_definition": w["definition"], "hash_mode": w["hash_mode"], "stage_a_match": getattr(job, "match", None),
                       "stage_a_sketched_ks": out.get("sketched_ks"),
                       "parallelism": ("reads + alignment records sharded x%d; every rank the whole table and its k-mer index, the ranks' counters "
                                       "all-gathered (two bits a pair) and summed, the count lists streamed in shares" % world)
                       if getattr(job, "match", None) == "kmer" else
                       ("reads + alignment records sharded x%d, read sketches and sketch tables sharded by hash range" % world)},
            "roofline": roof,
            "kernels": kern,
            "kernel_ms_per_pass": kernels_ms,
            "kernel_note": "HIP-event time per kernel family from extra untimed steps with everything on ONE stream (nothing "
                           "overlaps); the timed passes themselves are pipelined over three streams",
            "sanity": {"top_genomes_recovered": out.get("top_ok"), "tot_rds": out.get("tot_rds"),
                       "sketch_sizes": out.get("sketch_sizes")},
        }
        # a dense table (>= 5 % of all k-mers pass its threshold: configs[3]'s 5 kb genomes) gets a resident index at load
        # (DESIGN.md §4): stage A then counts in it, and the sketch is the exact intersection with the table
        resident = sum(f.resident_bytes for f in getattr(job.engine, "filters", []) if f is not None)
        if knobs:
            res["config"]["knobs"] = knobs
        res["config"]["stage_a_tables"] = ("resident index of the genome table, %.1f GB in HBM" % (resident / 1e9)) if resident \
            else ("the index over the table's k-mers (minimizer gate, buckets of 32-byte entries) + one 32-bit counter per pair and pass"
                  if by_kmer else "counting tables per pass + the table's membership filter")
        tr = job.traffic_per_pass() if hasattr(job, "traffic_per_pass") else None
        if tr:
            # what rank 0 hands to the other ranks per pass, by collective (nothing at world size 1: `sketch_entries` x 12 B is
            # the all-to-all volume that W ranks split), and what it holds for stage A besides the batch
            res["exchange_bytes"] = dict(tr, note="bytes per pass rank 0 sends to other ranks; sketch entries are (hash u64, count "
                                                  "u32) = 12 B, the slice a rank keeps is not counted")
            res["resident_bytes"] = resident
        if world > 1:
            # the default workload differs between N = 1 (configs[2]) and N > 1 (configs[3] shapes): the figure this
            # line's per-GPU workload gives on ONE GPU (world 1, same collectives in the path), as committed
            ref = n1 if (n1 and "error" not in n1) else committed_run("config3_world1_forced_dist_bench.json", cfg)
            if ref:
                res["same_workload_n1"] = ref
                if n1 and "error" in n1:
                    res["same_workload_n1"]["note_in_run"] = n1["error"] + "; the committed figure is quoted instead"
            res["collective_selfcheck"] = "ok"
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a single-GPU-run figure
            res["cpu_baseline"], res["check"] = cpu_baseline_and_check(args, cfg, w, hip)
        if world == 1 and not args.no_definitions and not args.no_secondary:
            # the other definitions of stage A/B on the same reads, genomes and records, the same pipelined loop
            del job
            job = None
            try:
                res["definitions"] = other_definitions(hip, args, cfg, w)
                res["definitions"]["%s_mode%d" % (w["definition"], w["hash_mode"])] = {
                    "definition": w["definition"], "hash_mode": w["hash_mode"], "ms_per_pass": ms, "value": res["value"], "unit": "reads/s",
                    "steps": args.steps, "note": "the headline of this line"}
                # (as VERDICT r03 asked: the reference pipeline under both hash definitions by name)
                res["reference_pipeline"] = {"mode%d" % m: {k: res["definitions"]["reference_pipeline_mode%d" % m][k]
                                                             for k in ("ms_per_pass", "value", "unit")} for m in (0, 1)}
            except Exception as e:  # noqa: BLE001  (a secondary figure must not take the headline down)
                res["definitions"] = {"error": repr(e)}
        if not args.no_secondary and world == 1 and cfg["config"] != 1:
            job = None
            res["secondary"] = secondary_config1(hip, args)
            try:
                # the kept command line on THIS workload from files on disk (page cache): the same reads as a FASTQ file,
                # the same tables as a stored sketch table, 1.25 SAM lines per read
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import bench_cli
                res["with_ingest"] = bench_cli.measure(nreads, G=cfg["genomes"], ks=tuple(cfg["ks"]), glen=cfg["genome_len"],
                                                       sketch_n=args.sketch_n, workload=w)
            except BaseException as e:  # noqa: BLE001  (a secondary figure must not take the headline down — a sys.exit of the command line neither)
                res["with_ingest"] = {"error": repr(e)}
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
