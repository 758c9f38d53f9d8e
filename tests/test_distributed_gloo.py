"""CPU, world_size 2 over gloo: the sharding choreography of metalign_amd/distributed.py.

The compute calls are served by an ORACLE-backed engine here (test infrastructure); what is under test is
the exchange logic — sketch all-gather + merge, carried-state composition across shard edges, lookahead
record, and the single all-reduce — whose result must equal the single-process oracle on the unsharded input."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleEngine:
    """Same interface as metalign_amd.distributed.HipEngine, numpy + CPU tensors."""
    device = "cpu"

    def __init__(self, torch_mod):
        import oracle
        self.o = oracle
        self.torch = torch_mod

    supports_kmer = True  # stage A by k-mer identity (ShardJob(match="kmer")): this rank's reads against the WHOLE table, counters summed
    cs = 3                # the counters' saturation value (0: they do not saturate: the all-reduce instead of the two-bit all-gather)

    class _KSk:
        """What HipEngine's _KmerSketch is to a job: a sample's counters, no sketch."""
        class _Counters:  # (the handle a job passes back to kmer_pack / kmer_merge / kmer_raw: _hip.KmerCounts in the library's engine)
            def __init__(self, counts):
                self.counts = counts

        def __init__(self, counts):
            self.counts, self.size = self._Counters(counts), None

        def free(self):
            pass

    def kmer_saturation(self):
        return self.cs

    def kmer_pack(self, kc):
        c = np.minimum(kc.counts, 3).astype(np.uint32)
        c = np.concatenate([c, np.zeros((-len(c)) % 16, np.uint32)]).reshape(-1, 16)
        return self.torch.from_numpy((c << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1, dtype=np.uint32).view(np.int32).copy())

    def kmer_merge(self, kc, every):
        w = every.numpy().view(np.uint32)  # [ranks][words]
        fields = (w[:, :, None] >> (2 * np.arange(16, dtype=np.uint32))) & 3
        kc.counts = fields.sum(axis=0, dtype=np.uint32).reshape(-1)[: len(kc.counts)]

    def kmer_raw(self, kc):
        kc.counts = np.ascontiguousarray(kc.counts, dtype=np.uint32)
        return self.torch.from_numpy(kc.counts.view(np.int32))  # (shares its memory: the all-reduce lands in the counters)

    def load(self, rbases, roffsets, recs, has_lookahead, ref2tax, ntax, tables, reftable=None):
        self.kmer = reftable is not None and bool(getattr(self, "match_kmer", False))
        self.rb, self.ro = rbases, roffsets
        self.recs, self.has_look = recs, has_lookahead
        self.ref2tax, self.ntax = ref2tax, ntax
        self.tables = tables  # one (hashes, offsets) per k: this rank's hash-range slice
        self.reftable = reftable  # the reference pipeline: this rank's share of the table (host arrays)
        self.ncols = len(reftable["ks"]) if reftable is not None else len(tables)

    class _Sk:
        def __init__(self, h, c, trunc):
            self.h, self.c, self.truncated = h, c, trunc
            self.size = len(h)
            self.last_hash = int(h[-1]) if len(h) else 0
            self.bound = None

        def free(self):
            pass

    def split_sketch(self, sk, bounds):
        return [int(np.searchsorted(sk.h, np.uint64(b), side="left")) for b in bounds]

    def set_sketch_bound(self, sk, truncated, bound):
        sk.truncated, sk.bound = bool(truncated), int(bound)

    def sketch_local(self, ks, hmaxs, s):
        if getattr(self, "kmer", False):
            c, _ = self.o.refpipe_count_kmers(self.rb, self.ro, ks[-1], self.reftable["kmer_hi"], self.reftable["kmer_lo"], cs=self.cs)
            return [self._KSk(c)]
        out = []
        for ki, (k, hmax) in enumerate(zip(ks, hmaxs)):
            h, c, t, _ = self.o.sketch_reads(self.rb, self.ro, k, hmax=hmax, s=s)
            sk = self._Sk(h, c, t)
            sk.ki = ki
            out.append(sk)
        return out

    def export_sketch(self, sk):
        t = self.torch
        return t.from_numpy(sk.h.view(np.int64).copy()), t.from_numpy(sk.c.view(np.int32).copy())

    def merge_sketches(self, hashes_t, counts_t, k, s, any_truncated, bound, hash_range=None):
        h = hashes_t.numpy().view(np.uint64)
        c = counts_t.numpy().view(np.uint32).astype(np.uint64)
        order = np.argsort(h, kind="stable")
        h, c = h[order], c[order]
        uh, start = np.unique(h, return_index=True)
        # counters saturate at cs (the sketch definition): min(sum of min(c_r, cs), cs) == min(sum of c_r, cs)
        uc = np.minimum(np.add.reduceat(c, start), self.o.DEFAULT_CS).astype(np.uint32) if len(h) else np.zeros(0, np.uint32)
        trunc = bool(any_truncated)
        if any_truncated:
            keep = uh <= np.uint64(bound)
            uh, uc = uh[keep], uc[keep]
        if s and len(uh) > s:
            uh, uc, trunc = uh[:s], uc[:s], True
        return self._Sk(uh, uc, trunc)

    def containment(self, sks, ci):
        if isinstance(sks[0], self._KSk):  # the counters are the sample's by now (ShardJob._sum_kmer_counts): every column, on every rank
            return self.o.refpipe_containment_counts(sks[0].counts.counts, ci, self.reftable)
        if self.reftable is not None:
            # the reference pipeline: mark from this rank's pairs, the job's OR exchange, count over this rank's count-list runs
            share, sk = self.reftable, sks[0]
            G = share["ngenomes"]
            matched = self.o.refpipe_matched(sk.h, sk.c, ci, share["pair_hash"])
            marks = [self.torch.from_numpy(self.o.refpipe_mark_words(matched, t).view(np.int32)) for t in share["small"]]
            hook = getattr(self, "mark_exchange", None)
            ored = hook(marks) if hook is not None else marks
            hits = [self.o.refpipe_count_words(ored[ki].numpy().view(np.uint32), t, G) for ki, t in enumerate(share["small"])]
            hits.append(np.bincount(np.asarray(share["pair_gen"])[matched != 0], minlength=G).astype(np.uint32)[:G])
            sizes = [np.asarray(t["gsize"], dtype=np.uint32) for t in share["small"]] + [np.asarray(share["gsize"], dtype=np.uint32)]
            return np.stack(hits), np.stack(sizes)
        res = [self._containment_one(sk, ci, *self.tables[ki]) for ki, sk in enumerate(sks)]
        return np.stack([r[0] for r in res]), np.stack([r[1] for r in res])

    def _containment_one(self, sk, ci, dbh, dbo):
        if sk.bound is None:
            return self.o.containment(sk.h, sk.c, sk.truncated, ci, dbh, dbo)
        # slice of a sample sketch: complete up to the SAMPLE's last hash.  Express that for the oracle by
        # appending the bound as a sentinel entry with count 0 (never a hit) so that it is the "last hash".
        if sk.truncated:
            h = np.concatenate([sk.h[sk.h < np.uint64(sk.bound)], [np.uint64(sk.bound)]])
            c = np.concatenate([sk.c[sk.h < np.uint64(sk.bound)],
                                sk.c[sk.h == np.uint64(sk.bound)] if (sk.h == np.uint64(sk.bound)).any() else [np.uint32(0)]])
            return self.o.containment(h, c.astype(np.uint32), True, ci, dbh, dbo)
        return self.o.containment(sk.h, sk.c, False, ci, dbh, dbo)

    def profile_begin(self, pct_id, need_map=True):
        import shard_ref
        self.pct_id = pct_id
        self.nrecs = len(self.recs) - (1 if self.has_look else 0)
        outs = [shard_ref.run_shard(self.recs, self.nrecs, self.has_look, self.ref2tax, self.ntax, pct_id, x, False,
                                    0)["outgoing"] for x in (0, 1)]
        ngroups = int(((self.recs["ref_new"][: self.nrecs] >> 31) & 1).sum())
        return (outs[0], outs[1]), ngroups

    def profile_commit(self, incoming, first_shard, group_base, want_multimapped=True):
        import shard_ref
        r = shard_ref.run_shard(self.recs, self.nrecs, self.has_look, self.ref2tax, self.ntax, self.pct_id, incoming,
                                first_shard, group_base)
        scal = np.array([r["groups"], r["ambig"]], dtype=np.uint64)
        return r["count"], r["bases"], r["first_seen"], scal, r["mm"]


class PipelinedOracleEngine(OracleEngine):
    """OracleEngine + the x_* methods of the four-passes-in-flight schedule (ShardJob._run_exchange_pipelined), served on
    the host: what is under test is the schedule — words layout, the order of the collectives across ranks with several
    passes in flight, slot reuse, the repeat-the-all-gather path — not the compute."""
    force_stale_words = False  # publish an overflow count once: every rank must then repeat the all-gather

    def x_setup(self, W, G, T, bounds, nslot):
        K, C = len(bounds), self.ncols  # sketched k (words), columns (reduce buffer)
        self._xW, self._xNW, self._xnred, self._xbounds_list = W, K * (W + 4) + 3, 2 * C * G + 2 * T + W * T + C + 2, [list(b) for b in bounds]

    def x_begin(self):
        pass

    def x_end(self):
        pass

    def x_front(self, ks, hmaxs, s, pct_id):
        sks = self.sketch_local(ks, hmaxs, s)
        maps, ngroups = self.profile_begin(pct_id)
        return dict(sks=sks, maps=maps, ngroups=ngroups)

    def _word(self, P, overflow):
        W, word = self._xW, []
        if getattr(self, "kmer", False):  # (no sketch to cut into slices)
            return [0] * (len(self._xbounds_list) * (W + 4)) + [P["maps"][0], P["maps"][1], P["ngroups"]]
        for ki, sk in enumerate(P["sks"]):
            cuts = [0] + self.split_sketch(sk, self._xbounds_list[ki][1:W]) + [sk.size]
            last = sk.last_hash
            word += ([cuts[q + 1] - cuts[q] for q in range(W)]
                     + [int(sk.truncated), last - (1 << 64) if last >= (1 << 63) else last, sk.size, overflow])
        return word + [P["maps"][0], P["maps"][1], P["ngroups"]]

    def x_words(self, P, slot):
        stale = 1 if (self.force_stale_words and not getattr(self, "_staled", False)) else 0
        self._staled = True
        word = self._word(P, stale)
        if stale:  # a stale word really is wrong: all slice sizes but the first zeroed
            word[1:self._xW] = [0] * (self._xW - 1)
        return self.torch.as_tensor(np.asarray(word, dtype=np.int64))

    def x_redo_words(self, P, bounds, tail):
        return self.torch.as_tensor(np.asarray(self._word(P, 0), dtype=np.int64))

    def x_fetch_words(self, P, words_t, hold):
        P["words"] = words_t

    def x_wait_words(self, P):
        return P["words"].numpy().reshape(self._xW, self._xNW).tolist()

    def x_commit(self, P, incoming, first_shard, group_base):
        P["commit"] = (incoming, first_shard, group_base)

    def x_merge(self, P, rh, rc, k, lo, hi, any_trunc, bound):
        return self.merge_sketches(rh, rc, k, 0, any_trunc, bound, (lo, hi))

    def x_stage_b(self, P, merged, ci):
        P["merged"] = merged
        P["hs"] = self.containment(merged, ci)

    def x_collect(self, P, ci, want_multimapped):
        count, bases, first, scal, mm = self.profile_commit(*P["commit"], want_multimapped)
        hits, sizes = P["hs"]
        return hits, sizes, count, bases, first, scal, mm, [m.size for m in P["merged"]]  # (None for counters: ShardJob._fill_reduce)

    def x_reduce_buffer(self, P):
        P["buf"] = np.zeros(self._xnred, dtype=np.int64)
        return P["buf"]

    def x_reduce_tensor(self, P):
        return self.torch.from_numpy(P["buf"])

    def x_fetch_reduced(self, P, tb):
        P["red"] = tb

    def x_wait_reduced(self, P):
        return P["red"].numpy().copy()


def _worker(rank, world, port, tmpdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    import util
    from metalign_amd import distributed as mgd
    rng = np.random.default_rng(5)
    gb, go = util.random_genomes(rng, 10, 4000)
    n = 200
    tables = {k: oracle.sketch_genomes(gb, go, k, n) for k in (15, 21, 31)}
    rb, ro, src = util.sample_reads(rng, gb, go, 1200, 100, err=0.01, present=[2, 6])
    nref, ntax = 30, 9
    ref2tax = rng.integers(0, ntax, size=nref).astype(np.uint32)
    nrec = 5000
    recs = np.zeros(nrec, dtype=oracle.REC_DTYPE)
    new = rng.random(nrec) < 0.7
    new[0] = True
    recs["ref_new"] = rng.integers(1, nref, size=nrec).astype(np.uint32) | (new.astype(np.uint32) << 31)
    recs["total"] = 100
    recs["matched"] = rng.integers(30, 101, size=nrec)
    recs["flag_len"] = rng.choice([0, 16, 256, 272], size=nrec).astype(np.uint32) | (np.where(rng.random(nrec) < 0.8, 100, 0).astype(np.uint32) << 12)
    # shards: reads and records cut into `world` contiguous parts (records on read boundaries)
    starts = np.nonzero(new)[0]
    rcuts = [0] + [int(starts[len(starts) * i // world]) for i in range(1, world)] + [nrec]
    ncuts = [1200 * i // world for i in range(world + 1)]
    my_reads = (rb[int(ro[ncuts[rank]]): int(ro[ncuts[rank + 1]])], ro[ncuts[rank]: ncuts[rank + 1] + 1] - ro[ncuts[rank]])
    my_recs = recs[rcuts[rank]: rcuts[rank + 1]]
    want = oracle.profile_assign(recs, ref2tax, ntax, 0.5)
    res = {}
    # (k, s): one k given bare (the single-k surface), and several k in one pass (the reference's query is multi-k)
    for case, (kspec, s) in enumerate([(21, 0), ([15, 21, 31], 0), (21, 400), ([21, 31], 37)]):
        ks = [kspec] if np.isscalar(kspec) else list(kspec)
        dbh = tables[kspec][0] if np.isscalar(kspec) else [tables[k][0] for k in ks]
        dbo = tables[kspec][1] if np.isscalar(kspec) else [tables[k][1] for k in ks]
        job = mgd.ShardJob(None, dist, rank, world, k=kspec, ci=2, pct_id=0.5, s=s, engine=OracleEngine(torch))
        job.load(my_reads[0], my_reads[1], my_recs, ref2tax, dbh, dbo, ntax=ntax)
        out = job.step()
        # single-process truth
        checks = dict(count=np.array_equal(out["count"], want["count"]), bases=np.array_equal(out["bases"], want["bases"]),
                      first=np.array_equal(out["first_seen"], want["first_seen"]),
                      tot=out["tot_rds"] == want["tot_rds"], ambig=out["n_ambig"] == want["n_ambig"],
                      shape=out["hits_k"].shape == (len(ks), 10))
        for ki, k in enumerate(ks):
            th, to = tables[k]
            qh, qc, tr, _ = oracle.sketch_reads(rb, ro, k, hmax=int(th.max()), s=s)
            hits, sizes = oracle.containment(qh, qc, tr, 2, th, to)
            checks["hits%d" % k] = np.array_equal(out["hits_k"][ki], hits)
            checks["sizes%d" % k] = np.array_equal(out["sizes_k"][ki], sizes)
            checks["qn%d" % k] = out["sketch_sizes"][ki] == len(qh)
        checks["last_k_alias"] = np.array_equal(out["hits"], out["hits_k"][-1]) and out["sketch_size"] == out["sketch_sizes"][-1]
        # the four-passes-in-flight schedule (run()) must give the same sample-wide results, also when a rank's words
        # turn out stale and the all-gather is repeated
        pe = PipelinedOracleEngine(torch)
        pe.force_stale_words = (s == 0 and rank == world - 1)
        pjob = mgd.ShardJob(None, dist, rank, world, k=kspec, ci=2, pct_id=0.5, s=s, engine=pe)
        pjob.load(my_reads[0], my_reads[1], my_recs, ref2tax, dbh, dbo, ntax=ntax)
        pout = pjob.run(5)
        for key in ("hits_k", "sizes_k", "count", "bases", "first_seen"):
            checks["run_" + key] = np.array_equal(pout[key], out[key])
        checks["run_scalars"] = (pout["tot_rds"], pout["n_ambig"], pout["sketch_sizes"]) == (out["tot_rds"], out["n_ambig"], out["sketch_sizes"])
        if s == 0:
            checks["run_repeated_gather"] = getattr(pjob, "words_redone", 0) == 1
        ok = all(checks.values())
        if not ok:
            print("rank", rank, "case", case, "FAILED:", [k for k, v in checks.items() if not v], flush=True)
        res[case] = bool(ok)
    # ---- the reference pipeline (reads sketched at the largest k only; the prefix bitmaps OR-ed across the ranks) ----
    for case, (ks, mode) in enumerate([([15, 21, 31], 0), ([21], 0), ([9, 12, 15, 21], 1)], start=100):
        oracle.set_hash_mode(mode)
        h, khi, klo, o = oracle.sketch_genomes_kmers(gb, go, ks[-1], n)
        full = oracle.refpipe_build(h, khi, klo, o, ks)
        qh, qc, _, _ = oracle.sketch_reads(rb, ro, ks[-1], hmax=int(h.max()))
        whits, wsizes = oracle.refpipe_containment(qh, qc, 2, full)
        job = mgd.ShardJob(None, dist, rank, world, k=ks, ci=2, pct_id=0.5, engine=OracleEngine(torch), definition="reference_pipeline", match="hash")
        job.load(my_reads[0], my_reads[1], my_recs, ref2tax, full, ntax=ntax)
        out = job.step()
        checks = dict(hits=np.array_equal(out["hits_k"], whits), sizes=np.array_equal(out["sizes_k"], wsizes),
                      qn=out["sketch_sizes"] == [len(qh)], count=np.array_equal(out["count"], want["count"]),
                      first=np.array_equal(out["first_seen"], want["first_seen"]), some=bool(whits.any()))
        pjob = mgd.ShardJob(None, dist, rank, world, k=ks, ci=2, pct_id=0.5, engine=PipelinedOracleEngine(torch), definition="reference_pipeline", match="hash")
        pjob.load(my_reads[0], my_reads[1], my_recs, ref2tax, full, ntax=ntax)
        pout = pjob.run(4)
        for key in ("hits_k", "sizes_k", "count", "bases", "first_seen"):
            checks["run_" + key] = np.array_equal(pout[key], out[key])
        # the bytes a rank reports having sent per pass: every sketch entry routed once (12 B for those that leave the rank),
        # one all-reduce of the results, the words of the all-gather
        tr, ptr = job.traffic_per_pass(), pjob.traffic_per_pass()
        mine = int(oracle.sketch_reads(my_reads[0], my_reads[1], ks[-1], hmax=int(h.max()))[0].size)
        checks["traffic_passes"] = tr["passes"] == 1 and ptr["passes"] == 4
        checks["traffic_entries"] = tr["sketch_entries"] == mine and ptr["sketch_entries"] == mine
        checks["traffic_all_to_all"] = 0 < tr["sketch_all_to_all"] <= 12 * mine and tr["sketch_all_to_all"] % 12 == 0
        checks["traffic_all_reduce"] = tr["results_all_reduce"] > 0 and tr["words_all_gather"] > 0
        checks["traffic_marks"] = (tr["prefix_marks_all_to_all"] == 0) == (len(ks) == 1)
        checks["traffic_total"] = tr["total_bytes"] == sum(tr[k] for k in ("words_all_gather", "sketch_all_to_all", "prefix_marks_all_to_all", "results_all_reduce"))
        ok = all(checks.values())
        if not ok:
            print("rank", rank, "refpipe case", case, "FAILED:", [k for k, v in checks.items() if not v], flush=True)
        res[case] = bool(ok)
        # ... and with k-mers met BY IDENTITY: every rank the whole table and its own reads, the ranks' counters summed — two bits a
        # pair through one all-gather (cs = 3) or the counters themselves through an all-reduce (cs = 0) — on every rank every column
        if ks[-1] >= 15:
            for cs in (3, 0):
                checks = {}
                for label, eng_cls, passes in (("step", OracleEngine, 0), ("run", PipelinedOracleEngine, 3)):
                    eng = eng_cls(torch)
                    eng.cs = cs
                    kjob = mgd.ShardJob(None, dist, rank, world, k=ks, ci=2, pct_id=0.5, engine=eng, definition="reference_pipeline", match="kmer")
                    kjob.load(my_reads[0], my_reads[1], my_recs, ref2tax, full, ntax=ntax)
                    kout = kjob.run(passes) if passes else kjob.step()
                    checks[label + "_match"] = kjob.match == "kmer" and kout["match"] == "kmer"
                    checks[label + "_hits"] = np.array_equal(kout["hits_k"], whits) and np.array_equal(kout["sizes_k"], wsizes)
                    checks[label + "_qn"] = kout["sketch_sizes"] == [int(whits[-1].sum())]
                    checks[label + "_stage_c"] = np.array_equal(kout["count"], want["count"]) and np.array_equal(kout["first_seen"], want["first_seen"])
                    ktr = kjob.traffic_per_pass()
                    npairs = len(full["pair_hash"])
                    checks[label + "_traffic"] = ktr["kmer_counts_all_gather"] == ((world - 1) * 4 * ((npairs + 15) // 16) if cs else 4 * npairs) and \
                        ktr["sketch_all_to_all"] == 0 and ktr["prefix_marks_all_to_all"] == 0
                if not all(checks.values()):
                    print("rank", rank, "refpipe-by-identity case", case, "cs", cs, "FAILED:", [k for k, v in checks.items() if not v], flush=True)
                res[(case, "kmer", cs)] = bool(all(checks.values()))
        oracle.set_hash_mode(0)
    with open(os.path.join(tmpdir, "rank%d.txt" % rank), "w") as fh:
        fh.write(repr(res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharding_matches_single_process(tmp_path, world):
    import torch.multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        want = {0: True, 1: True, 2: True, 3: True}
        for case in (100, 101, 102):  # the reference pipeline by hash value, then by identity with both ways of summing the counters
            want.update({case: True, (case, "kmer", 3): True, (case, "kmer", 0): True})
        assert (tmp_path / ("rank%d.txt" % r)).read_text() == repr(want)


def _sam_worker(rank, world, port, tmpdir):
    """One rank of map_and_profile's multi-GPU launch, on the CPU over gloo: its line-aligned byte range of the SAM file
    tokenised (host tokeniser here), then map_and_profile.gather_record_pieces — the all-gathers, send / recv, the
    new-read bits at the cuts — exactly as map_and_process_file_dist calls it with device tensors under RCCL."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from metalign_amd import map_and_profile as mp
    dist.init_process_group("gloo", rank=rank, world_size=world)
    path = os.path.join(tmpdir, "a.sam")
    text = open(path).read()
    idx = {"ACC%03d.1" % i: i for i in range(20)}
    res = {}
    for case in ("records", "one rank cannot tokenise its piece"):
        a, b = mp.sam_range_of_rank(path, rank, world)
        tk = mp._Tokeniser(idx)
        for ln in text[a:b].splitlines(True):
            tk.feed(ln)
        recs = tk.records()
        n = len(recs)
        mine = torch.from_numpy(np.ascontiguousarray(recs).view(np.int32).copy()) if n else None
        bad = int(case != "records" and rank == world - 1)
        got = mp.gather_record_pieces(torch, dist, rank, world, mine, n, (mp.first_retained_qname(path, a, b) or "") if n else "",
                                      tk.prev if n else "", bad, "cpu")
        if case != "records":
            res[case] = got is None if rank == 0 else got == "done"
        elif rank != 0:
            res[case] = got == "done"
        else:
            buf, total = got
            whole = mp.tokenise_sam(text.splitlines(True), idx)
            res[case] = total == len(whole) and np.array_equal(buf[: 4 * total].numpy().view(whole.dtype), whole)
    with open(os.path.join(tmpdir, "sam_rank%d.txt" % rank), "w") as fh:
        fh.write(repr(res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 5])
def test_sam_record_pieces_gathered_over_gloo(tmp_path, world):
    import torch.multiprocessing as mp
    import socket
    rng = np.random.default_rng(world)
    lines = ["@HD\tVN:1.6"]
    for r in range(300):
        for j in range(int(rng.integers(1, 5))):
            seq = "ACGT" * int(rng.integers(5, 30)) if j == 0 else "*"
            lines.append("\t".join(["q%d" % r, "0" if j == 0 else "256", "ACC%03d.1" % int(rng.integers(0, 20)), "1", "60",
                                     "%dM" % (len(seq) if j == 0 else 40), "*", "0", "0", seq, "I" * len(seq) if j == 0 else "*",
                                     "NM:i:0"]))
    (tmp_path / "a.sam").write_text("\n".join(lines) + "\n")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_sam_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / ("sam_rank%d.txt" % r)).read_text() == repr({"records": True, "one rank cannot tokenise its piece": True})


def test_shard_reference_equals_c_oracle_unsharded():
    """The shard-aware Python restatement used by the engine above == the pinned C oracle on whole streams."""
    import oracle
    import shard_ref
    rng = np.random.default_rng(17)
    for trial in range(4):
        n, nref, ntax = 3000, 25, 7
        ref2tax = rng.integers(0, ntax, size=nref).astype(np.uint32)
        recs = np.zeros(n, dtype=oracle.REC_DTYPE)
        new = rng.random(n) < (0.5 + 0.1 * trial)
        new[0] = True
        recs["ref_new"] = rng.integers(0, nref, size=n).astype(np.uint32) | (new.astype(np.uint32) << 31)
        recs["total"] = 100
        recs["matched"] = rng.integers(20, 101, size=n)
        recs["flag_len"] = rng.choice([0, 16, 256, 2048, 99, 147, 355, 403, 73, 137], size=n).astype(np.uint32) | (
            np.where(rng.random(n) < 0.7, 100, 0).astype(np.uint32) << 12)
        want = oracle.profile_assign(recs, ref2tax, ntax, 0.5)
        got = shard_ref.run_shard(recs, n, False, ref2tax, ntax, 0.5, 1, True, 0)
        assert np.array_equal(got["count"], want["count"]) and np.array_equal(got["bases"], want["bases"])
        assert np.array_equal(got["first_seen"], want["first_seen"])
        assert (got["groups"], got["ambig"]) == (want["tot_rds"], want["n_ambig"])
        assert [m[0] for m in got["mm"]] == list(want["mm_read"])
        flat = [t for m in got["mm"] for t in m[1]]
        assert flat == list(want["mm_tax"]) and [m[2] for m in got["mm"]] == list(want["mm_hitlen"])


def test_compose_incoming():
    from metalign_amd.distributed import compose_incoming
    ident, const0, const1, flip = (0, 1), (0, 0), (1, 1), (1, 0)
    assert compose_incoming([ident, ident, ident], 2) == 1
    assert compose_incoming([const0, ident], 2) == 0
    assert compose_incoming([const0, flip], 2) == 1
    assert compose_incoming([const1, const0, ident], 3) == 0
    assert compose_incoming([flip], 0) == 1
