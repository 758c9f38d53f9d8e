"""One process per GPU: sharding of the hot path and its two exchange steps.

Partitioning
  reads / alignment records   contiguous shards in stream order, cut on read boundaries (rank r holds shard r)
  genome sketch table         sharded by HASH RANGE: rank r holds, for every genome, the part of its sketch that
                              falls in [ b[r], b[r+1] ), b = the r/W quantiles of the table's hashes (table_bounds:
                              equal-sized slices), and a genome's containment is the SUM of its per-slice hit counts
Exchanges (torch.distributed; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests)
  1. all-to-all of the per-rank read sketches by hash range (each rank sends 1/W of its sketch to every peer; the
     bytes a rank moves do not grow with W) -> every rank merges what it receives into ITS slice of the sample
     sketch (counts of equal hashes add; bottom-s is taken over the rank-ordered slices).
  2. all-gather of one tiny word per rank: the shard's 2-bit carried-state map and its read count, so each
     rank can compose the state entering its shard (scripts/map_and_profile.py:229-232 crosses shard edges).
  3. ONE all-reduce(sum, int64) of [containment hits | sizes | per-taxon read counts | bases |
     per-rank first-seen slots | tot_rds, n_ambig] -- hits and sizes are true sums over the hash slices.
     <= a few MB: latency-bound on xGMI, not bandwidth-bound.
Several k (the reference's query is multi-k: `30-60-10`, scripts/select_db.py:75): every k has its own sketch table,
membership filter, hash-range bounds and sample sketch; ONE pass hashes the reads for all of them (stage A, fused
into one launch when the library has the k set instantiated), runs stage B per k and stage C once.  The exchanges
carry all k together: the per-rank words of every k travel in the one all-gather, and the one all-reduce holds
[hits | sizes] for every k; the sketch slices move in one all-to-all round PER k (two all_to_all_single calls each:
hashes and counts), all of them queued before the first is waited for.
The compute calls go through an `engine` (HipEngine below: libmetalign_hip.so on this rank's GPU).  The
CPU tests substitute an oracle-backed engine to check the choreography under gloo; product code never does.
"""
import os

import numpy as np

from . import _hip

U64_MAX = _hip.U64_MAX


def _u64(x):
    """int64 tensor element (two's complement) -> python int in [0, 2^64)."""
    return int(x) & 0xFFFFFFFFFFFFFFFF


def slice_bounds(hmax, world):
    """Equal-WIDTH hash-range slices: rank r owns hashes in [b[r], b[r+1]); b[0] = 0, b[world] = hmax + 1."""
    return [((int(hmax) + 1) * r) // world for r in range(world + 1)]


def table_bounds(table_hashes, world, hmax=None, presorted=False):
    """Equal-LOAD hash-range slices of a sketch table: b[r] = the r/world quantile of the table's hashes, b[0] = 0,
    b[world] = hmax + 1.  A bottom-n sketch of genome g is uniform on [0, its own n-th smallest hash], and hmax is the
    MAXIMUM of those over all genomes, so the table thins out towards hmax: with equal-width slices the top one held
    3 M of a 200k-genome table's 200 M hashes and the others 28 M each (stage B and the sketch merge of seven ranks
    12 % over the mean, one rank idle).  Quantiles make the slices equal-sized whatever the genome sizes; every rank
    derives the same bounds from the same table."""
    h = np.asarray(table_hashes)
    total = len(h)
    if hmax is None:
        hmax = int(h.max()) if total else 0
    if world <= 1:
        return [0, int(hmax) + 1]
    if total == 0:
        return [0] + [int(hmax) + 1] * world
    kth = [total * r // world for r in range(1, world)]
    cut = h[kth] if presorted else np.partition(h, kth)[kth]
    b = [0] + [int(x) for x in cut] + [int(hmax) + 1]
    for r in range(1, world + 1):  # never decreasing (tiny tables with repeated hashes)
        b[r] = max(b[r], b[r - 1])
    return b


def table_slice(dbh, dbo, lo, hi):
    """The part of every genome sketch that falls in [lo, hi): (hashes, offsets[G+1]); order preserved."""
    dbh = np.asarray(dbh)
    keep = (dbh >= np.uint64(lo)) & (dbh < np.uint64(hi)) if hi <= U64_MAX else (dbh >= np.uint64(lo))
    csum = np.zeros(len(dbh) + 1, dtype=np.uint64)
    np.cumsum(keep, out=csum[1:])
    return dbh[keep], csum[np.asarray(dbo, dtype=np.int64)]


def compose_incoming(maps, rank):
    """State entering shard `rank`: start from 1 (the phantom first boundary, :155-156) and apply the maps of
    the preceding non-empty shards in order.  maps[r] = (out_if_in_0, out_if_in_1)."""
    x = 1
    for r in range(rank):
        x = maps[r][x]
    return x


class _CudaView:
    """Zero-copy view of a library-owned HBM range for torch (via __cuda_array_interface__)."""

    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


# Stage A by k-mer identity against the read sketch, one k alone (2M reads x 2k genomes, one MI355X, tools/kcount_probe.py, the round's last
# kernel): k = 21: 1.37 against 1.09 ms, 23: 1.35 / 1.13, 24: 1.19 / 1.05, 25: 1.08 / 1.14, 26: 0.97 / 1.22, 27: 0.92 / 1.16, 31: 0.72 / 1.16,
# 41: 0.61 / 1.42, 60: 0.52 / 1.46 — a window of k has k - 14 - 2 e candidates, and a read of 150 bases closes 300 / (k - 13 - 2 e)
# runs: below k = 25 the runs cost more than the hashes they save.
KMER_MATCH_MIN_K, KMER_MATCH_MAX_K = 15, 64   # what mg_kcount.hip is built for
KMER_MATCH_DEFAULT_FROM_K = 25                # ... and from where it is what a job does when nobody says


def kmer_match_by_default(kmax):
    return KMER_MATCH_DEFAULT_FROM_K <= int(kmax) <= KMER_MATCH_MAX_K


class _KmerSketch:
    """What stage A BY K-MER IDENTITY leaves for stage B (the reference pipeline's default since round 6): one sample's
    occurrence counters of the table's k_max-mers (_hip.KmerCounts) instead of a sketch of hashes.  Quacks like a _hip.Sketch
    where a pass touches it: nothing to settle, nothing that can overflow; freeing it hands the counters back to the engine."""

    def __init__(self, engine, counts):
        self.engine, self.counts = engine, counts
        self.size = None  # (no sketch: ShardJob._results reports the matched pairs of the largest k in its place)

    def resolve(self):
        return False

    def free(self):
        if self.counts is not None:
            # zeroed NOW, on the main stream (0.2 ms of copies and fills that run beside the next pass's counting), not in front of
            # the counting kernel on the stage-A stream, where they were a hole of that length between two launches
            self.counts.reset()
            self.engine._kc_free.append(self.counts)
            self.counts = None


class HipEngine:
    wg_per_cu_exchange = 3  # the hashing kernel's workgroups per CU beside the exchange chain (x_begin)

    """Compute side of one rank: device-resident inputs + calls into libmetalign_hip.so."""

    def __init__(self, hip, torch_mod=None):
        self.hip = hip
        self.torch = torch_mod  # None in single-process mode: results are returned as numpy arrays
        self.keep = []
        self.filters = []  # one membership pre-filter per k (set_filter), None = unfiltered

    # ---- inputs ----
    def load(self, rbases, roffsets, recs, has_lookahead, ref2tax, ntax, tables, reftable=None):
        """tables: one (hashes, offsets[G+1]) per k — this rank's hash-range slice of each sketch table — or, for a
        table already laid out hash-major on disk (formats.SketchTable.pairs), a dict(pair_hash=, pair_gen=, gsize=).
        reftable (the reference pipeline: `tables` is then empty): this rank's share of the table as a _hip.RefTable —
        stage A sketches the largest k only, stage B gives a column per k of the table."""
        hip = self.hip
        self.reftable = reftable
        self.nreads = len(roffsets) - 1
        self.d_rb = hip.array(rbases if len(rbases) else np.zeros(1, np.uint8))
        self.d_ro = hip.array(roffsets)
        self.nrecs = len(recs) - (1 if has_lookahead else 0)
        self.has_lookahead = has_lookahead
        self.d_recs = hip.array(recs if len(recs) else np.zeros(1, _hip.REC_DTYPE))
        self.d_r2t = hip.array(ref2tax)
        self.nref, self.ntax = len(ref2tax), ntax
        if reftable is not None:
            self.tables = [reftable.kmax_table()]
            self.nk = len(reftable.ks)  # columns of the result
        else:
            self.tables = [hip.upload_table_sorted(**t) if isinstance(t, dict) else hip.upload_table(t[0], t[1]) for t in tables]
            self.nk = len(tables)
        self.nsk = len(self.tables)  # k the reads are sketched at (the reference pipeline: the largest only)
        self.ngen_local = self.tables[0].ngenomes
        g = self.gpad = max(self.ngen_local, 1)
        # per k: [hits g | sizes g], all k in one buffer: one read-back
        self.d_hs = hip.empty(2 * g * self.nk, np.uint32)
        # [count T | bases T | first_seen T | scalars 2]
        self.d_acc = hip.empty(3 * ntax + 2, np.uint64)
        # page-locked landing buffers: a step queues both read-backs behind its kernels and syncs once
        self.h_hs = hip.pinned(2 * g * self.nk, np.uint32)
        self.h_acc = hip.pinned(3 * ntax + 2, np.uint64)
        self.filters += [None] * (self.nsk - len(self.filters))
        # stage A by k-mer identity (ShardJob(match="kmer")): the counters of the passes in flight, handed out and taken back
        self.kmer = reftable is not None and bool(getattr(self, "match_kmer", False))
        self._kc_free = []
        # (several ranks, each with the whole table: a rank streams its share of the count lists, the columns of k < k_max are summed)
        share = getattr(self, "count_share", None)
        self.kmer_count_sharded = bool(self.kmer and share and share[1] > 1)
        if reftable is not None and hasattr(reftable, "set_count_share"):
            reftable.set_count_share(*(share if self.kmer_count_sharded else (0, 1)))  # (the handle remembers: a job says it every time)
        if self.kmer and not reftable.has_kmer_index:
            raise _hip.HipError("the reference-pipeline table has no k-mer index (RefTable.index_kmers)")
        if self.kmer:
            self.reserve_counters(3)

    def reserve_counters(self, n):
        """Stage A by k-mer identity: the sets of counters the passes in flight will hand around, made NOW — a set is gigabytes at a
        RefSeq-scale table (5.6 GB at 2 x 10^8 k-mers) and a hipMalloc of that size is a tenth of a second: made on demand they
        landed in the first passes of a run, sometimes inside a timed region (configs[3] shapes: 13 or 130 ms per pass, by luck)."""
        while len(self._kc_free) < n:
            self._kc_free.append(self.reftable.kmer_counts())

    # ---- stage A ----
    def set_filter(self, ki, table_hashes):
        """Membership pre-filter over ALL hashes of the genome table of k number ki (also on a rank that holds a slice
        of it): the read sketch keeps only hashes that may be in the table — the reference's `-f ...bf`
        (select_db.py:70,75)."""
        self.filters += [None] * (ki + 1 - len(self.filters))
        self.filters[ki] = self.hip.filter_build(table_hashes)
        if len(table_hashes) and self.wants_resident_index(int(np.max(table_hashes)), len(table_hashes)):
            # a table whose largest hash filters little: its hashes seeded once into a resident counting table — a candidate
            # then costs one random access instead of a home slot plus a filter word, and no table is cleared per pass
            # (False: the hashes crowd some range and the bit filter stays)
            self.filters[ki].make_resident(table_hashes, int(np.max(table_hashes)), self.resident_spread(len(table_hashes)))

    def resident_spread(self, nhashes):
        """1 (half the load: fewer candidates that have to look a slot further — 28.5 against 29.7 ms per 12.5M reads at
        200k genomes) when the indexes at twice the size (two copies: one per stream that sketches) still fit half of the free
        HBM, else 0."""
        e = os.environ.get("MG_RESIDENT_SPREAD", "auto")
        if e in ("0", "1", "2", "3"):
            return int(e)
        free, _, pooled = self.hip.mem_info()
        return 1 if 2 * 2 * 48 * nhashes * max(getattr(self, "nsk_hint", getattr(self, "nk", 1)), 1) <= (free + pooled) // 2 else 0

    def wants_resident_index(self, hmax, nhashes):
        """MG_RESIDENT_INDEX=1 / 0 forces it on / off; otherwise: when at least 5 % of all k-mers pass the table's threshold
        (genomes of a few kb, or a mixed table: stage A is then bound by its random look-ups, not by its hashing) and
        the indexes of all k (32 to 64 bytes per hash, once per stream that sketches: two for a pipelined job) fit half of
        the free HBM."""
        if getattr(self, "bottom_s", 0):
            return False  # (the library would not use it: bottom-s sketches are cut from what passes the FILTER)
        e = os.environ.get("MG_RESIDENT_INDEX", "auto")
        if e in ("0", "1"):
            return e == "1"
        hash_range = 9999999999971.0 if self.hip.hash_mode == 1 else 2.0 ** 64
        if (hmax + 1) / hash_range < 0.05:
            return False
        free, _, pooled = self.hip.mem_info()
        return 2 * 48 * nhashes * max(getattr(self, "nsk_hint", getattr(self, "nk", 1)), 1) <= (free + pooled) // 2

    def set_filter_bits(self, ki, bits):
        """The same from the bit array the table builder stored (formats.SketchTable.filter_bits)."""
        self.filters += [None] * (ki + 1 - len(self.filters))
        self.filters[ki] = self.hip.filter_from_bits(bits)

    def sketch_local(self, ks, hmaxs, s):
        sks = self.sketch_local_async(ks, hmaxs, s)
        for sk in sks:
            sk.resolve()
        return sks

    def prime(self, ks, hmaxs, s):
        """One stage-A pass over the resident reads, settled and dropped: the library sizes its counting tables and sketch
        buffers from the distinct-to-candidate ratio of the PREVIOUS batch of the same k and has none yet — without this,
        every pass a pipelined job queues before its first read-back is sized for the worst case (a dense 200k-genome
        table: 13 GB of table and 5 GB of sketch per k and pass in flight, against 1-2 GB once the ratio is known)."""
        # (on the first stage-A stream, where a pipelined job's passes run: a table's resident index keeps one copy per
        # stream that has sketched with it, 8 GB per k at 200k genomes — the main stream need not own one)
        import time

        def one_pass():
            self.hip.sync()
            t0 = time.perf_counter()
            for sk in self.sketch_local(ks, hmaxs, s):
                sk.free()
            return time.perf_counter() - t0
        self.hip.stage_a_side_stream(True)
        try:
            one_pass()
            if any(f is not None and f.resident_bytes for f in self.filters[: len(ks)]) and \
                    os.environ.get("MG_RESIDENT_INDEX", "auto") != "1":
                # The index pays when most candidates ARE hashes of the table (a sample of the table's genomes, thresholds
                # near the genomes' own); when most are not — hash definition 1's prefix tables, where every k-mer is a
                # candidate and 2 % of a present genome's are members — the bit filter's one cached word rejects them for
                # less.  Which it is depends on the sample: measured, stage A once more each way (the caller drops the loser).
                with_index = one_pass()
                have = [f for f in self.filters[: len(ks)] if f is not None and f.resident_bytes]
                for f in have:
                    f.use_resident(False)
                try:
                    one_pass()  # (the first pass of this form sizes its tables for the worst case)
                    with_filter = one_pass()
                finally:
                    for f in have:
                        f.use_resident(True)
                return with_index, with_filter
        finally:
            self.hip.stage_a_side_stream(False)
        return None

    def drop_resident_indexes(self):
        for f in self.filters:
            if f is not None and f.resident_bytes:
                f.drop_resident()

    def sketch_local_async(self, ks, hmaxs, s):
        """The read sketches for every k, queued without a host sync (one fused launch when the library has the k set).
        match = "kmer": the reads' k_max-mers counted against the table's by identity instead (one launch, no sync)."""
        if getattr(self, "kmer", False):
            # (a set of counters comes zeroed: when it was made, or when it was handed back — the one handed back longest ago first)
            kc = self._kc_free.pop(0) if self._kc_free else self.reftable.kmer_counts()
            kc.add_dev(self.d_rb.ptr, self.d_ro.ptr, self.nreads, int(self.d_rb.count))
            return [_KmerSketch(self, kc)]
        return self.hip.sketch_reads_multi_dev_async(self.d_rb.ptr, self.d_ro.ptr, self.nreads, ks, hmaxs, s,
                                                     self.filters[: len(ks)])

    # ---- stage A by k-mer identity on several ranks: what ShardJob._sum_kmer_counts asks of an engine ----
    def kmer_saturation(self):
        return self.hip.count_saturation()

    def kmer_pack(self, kc):
        """min(counter, 3) of every pair, two bits each -> an int32 tensor on the device (mg_kcounts_pack2_dev)."""
        t = self.torch
        n32 = max(kc.pack2_bytes() // 4, 1)
        mine = t.empty(n32, dtype=t.int32, device="cuda")
        kc.pack2_dev(mine.data_ptr())
        return mine

    def kmer_merge(self, kc, every):
        """The counters := the sum over the ranks' packed arrays (rows of `every`)."""
        kc.merge2_dev(every.data_ptr(), int(every.shape[0]), 4 * int(every.shape[1]))

    def kmer_raw(self, kc):
        """The 32-bit counters as an int32 tensor (for an all-reduce in place), the main stream behind whoever wrote them last."""
        ptr, n = kc.device()
        kc.wait()
        return self.torch.as_tensor(_CudaView(ptr, max(n, 1), "<i4"), device="cuda")

    def export_sketch(self, sk):
        """(hashes int64 tensor, counts int32 tensor) on this rank's device, zero-copy."""
        t = self.torch
        n = sk.size
        if n == 0:
            return t.zeros(0, dtype=t.int64, device="cuda"), t.zeros(0, dtype=t.int32, device="cuda")
        ph, pc = sk.device_ptrs()
        return (t.as_tensor(_CudaView(ph, n, "<i8"), device="cuda"), t.as_tensor(_CudaView(pc, n, "<i4"), device="cuda"))

    def merge_sketches(self, hashes_t, counts_t, k, s, any_truncated, bound, hash_range=None):
        """hash_range = (lo, hi) inclusive when every hash is known to lie in it (a hash-range slice)."""
        n = int(hashes_t.numel())
        self.keep.append((hashes_t, counts_t))
        lo, hi = hash_range if hash_range is not None else (1, 0)
        return self.hip.sketch_merge_dev(hashes_t.data_ptr(), counts_t.data_ptr(), n, k, lo, hi, s, any_truncated, bound)

    def split_sketch(self, sk, bounds):
        return sk.split(bounds)

    # ---- stage B ----
    def _hs_ptrs(self, base_ptr, ki):
        g = self.gpad
        return base_ptr + 4 * (2 * g * ki), base_ptr + 4 * (2 * g * ki + g)

    def _stage_b(self, sks, ci, base_ptr):
        """Stage B of every k: one launch of each kernel for all of them (mg_containment_multi_dev; at most four k per call).
        The reference pipeline: the one sketch of the largest k against the table's pairs and count lists
        (mg_refpipe_containment_dev), a column per k."""
        if getattr(self, "reftable", None) is not None:
            ptrs = [self._hs_ptrs(base_ptr, ki) for ki in range(self.nk)]
            hook = getattr(self, "mark_exchange", None)
            if isinstance(sks[0], _KmerSketch):
                # (a rank of a multi-GPU job holds the whole table, and its counters the whole sample's sums by now: every rank
                # computes every column, ShardJob._fill_reduce lets rank 0's through)
                self.hip.refpipe_containment_counts_dev(sks[0].counts, self.reftable, ci, [p[0] for p in ptrs], [p[1] for p in ptrs])
                return
            if hook is None:
                self.hip.refpipe_containment_dev(sks[0], self.reftable, ci, [p[0] for p in ptrs], [p[1] for p in ptrs])
                return
            # a rank of a multi-GPU job: the matched pairs of ITS hash range mark prefixes anywhere in the table; the ranks'
            # bitmaps are OR-ed (the job's exchange: every rank receives its own prefix share of every other rank's bitmap)
            # before the rank's runs of the count lists are streamed against them
            t, last = self.torch, self.nk - 1
            self.hip.refpipe_mark_dev(sks[0], self.reftable, ci, ptrs[last][0], ptrs[last][1])
            mine = []
            for ki in range(last):
                ptr, nw = self.reftable.marks(ki)
                mine.append(t.as_tensor(_CudaView(ptr, nw, "<i4"), device="cuda") if nw else t.zeros(0, dtype=t.int32, device="cuda"))
            ored = hook(mine)
            self.hip.refpipe_count_dev(self.reftable, [o.data_ptr() if o.numel() else self.reftable.marks(ki)[0] for ki, o in enumerate(ored)],
                                       [p[0] for p in ptrs[:last]], [p[1] for p in ptrs[:last]])
            self._marks_keep = (mine, ored)  # (until the next call; the count is queued on the stream torch allocates for)
            return
        ptrs = [self._hs_ptrs(base_ptr, ki) for ki in range(len(sks))]
        for a in range(0, len(sks), 4):  # (kMaxContainK of mg_contain.hip)
            self.hip.containment_multi_dev(sks[a:a + 4], self.tables[a:a + 4], ci, [p[0] for p in ptrs[a:a + 4]],
                                           [p[1] for p in ptrs[a:a + 4]])

    def _stage_b_again(self, sks, ki, ci, base_ptr):
        """Stage B of sketch number ki once more (its counting table overflowed and it was rebuilt at resolution)."""
        self.hip.sync()
        if getattr(self, "reftable", None) is not None:
            self._stage_b(sks, ci, base_ptr)
        else:
            self.hip.containment_dev(sks[ki], self.tables[ki], ci, *self._hs_ptrs(base_ptr, ki))
        self.hip.sync()

    def _hs_split(self, hs):
        g, G = self.gpad, self.ngen_local
        hs = np.asarray(hs).reshape(self.nk, 2, g)
        return hs[:, 0, :G].copy(), hs[:, 1, :G].copy()

    def containment(self, sks, ci):
        """-> (hits[K][G], sizes[K][G]) of this rank's table slices."""
        self._stage_b(sks, ci, self.d_hs.ptr)
        return self._hs_split(self.d_hs.download())

    def containment_and_commit_results(self, sks, ci, want_multimapped):
        """Stage B's kernels, then BOTH read-backs (containment counts, stage-C accumulators of a commit queued
        earlier with profile_commit_launch) behind them: one synchronisation."""
        T = self.ntax
        # the per-genome counts (8 B per genome) are written by the kernel straight into page-locked host memory
        self._stage_b(sks, ci, self.h_hs.ptr)
        self.hip.stage_c_join()  # the accumulators are read on the main stream: it waits for stage C's stream here
        self.h_acc.fetch_async(self.d_acc.ptr)
        self.hip.sync()
        for ki, sk in enumerate(sks):
            if sk.resolve():  # stage A's counting table overflowed and the sketch was rebuilt: stage B again
                self._stage_b_again(sks, ki, ci, self.h_hs.ptr)
        hs, acc = self._hs_split(self.h_hs.array), self.h_acc.array.copy()
        mm = self.shard.multimapped() if want_multimapped else None
        self.shard.free()
        return hs, (acc[:T], acc[T:2 * T], acc[2 * T:3 * T], acc[3 * T:], mm)

    # ---- stage C ----
    def set_sketch_bound(self, sk, truncated, bound):
        sk.set_bound(truncated, bound)

    def profile_begin(self, pct_id, need_map=True):
        """Pass A of stage C.  The composed state map / read count are only read back (one stream sync) when a
        neighbouring shard needs them."""
        self.shard = self.hip.profile_begin_dev(self.d_recs.ptr, self.nrecs, self.has_lookahead, self.d_r2t.ptr,
                                                self.nref, self.ntax, pct_id)
        if not need_map:
            return (0, 1), 0
        return self.shard.state_map(), self.shard.ngroups

    def profile_begin_async(self, pct_id):
        """Pass A of stage C queued without a sync; profile_map() fetches its two words later."""
        self.shard = self.new_shard_async(pct_id)

    def new_shard_async(self, pct_id):
        """A stage-C handle over the resident records with its map-only pass already queued (a pipelined job starts
        the NEXT pass's handle while the current pass is still being finished)."""
        shard = self.hip.profile_begin_dev(self.d_recs.ptr, self.nrecs, self.has_lookahead, self.d_r2t.ptr,
                                           self.nref, self.ntax, pct_id)
        shard.map_launch()
        return shard

    def profile_map(self):
        return self.shard.state_map(), self.shard.ngroups

    # ---- queue-ahead passes (single shard): everything of a pass is queued, a marker is recorded behind it, and
    # the pass is read back later — after the NEXT pass has been queued, so the GPU runs pass after pass without
    # waiting for the host in between.  Two result sets alternate.
    def _result_sets(self):
        if not hasattr(self, "_sets"):
            hip, g, T = self.hip, self.gpad, self.ntax
            self._sets = [dict(d_acc=self.d_acc, h_acc=self.h_acc, h_hs=self.h_hs, ev=hip.event()),
                          dict(d_acc=hip.empty(3 * T + 2, np.uint64), h_acc=hip.pinned(3 * T + 2, np.uint64),
                               h_hs=hip.pinned(2 * g * self.nk, np.uint32), ev=hip.event())]
        return self._sets

    def queue_pass(self, slot, ks, hmaxs, s, ci, pct_id, side=False):
        rs = self._result_sets()[slot]
        T = self.ntax
        if side:  # stage A of consecutive passes on alternating streams: pass i+1's overlaps pass i's tail
            # (3 / 4: the two stage-A streams at the lowest priority — no collective in this schedule, and the short kernels of stage B and C
            # then get in ahead of the next pass's persistent stage-A kernel: 2.33 -> 2.22 ms per pass at configs[2])
            # (by hash value the sketch's own small kernels — sort, pack — ride the stage-A streams: default priority, as measured: 5.4 against 5.6 ms)
            self.hip.stage_a_side_stream((3 if getattr(self, "kmer", False) else 1) + (slot & 1))
        sks = self.sketch_local_async(ks, hmaxs, s)                                               # stage A
        shard = self.hip.profile_begin_dev(self.d_recs.ptr, self.nrecs, self.has_lookahead, self.d_r2t.ptr,
                                           self.nref, self.ntax, pct_id)
        base = rs["d_acc"].ptr
        shard.commit(True, True, 0, base, base + T * 8, base + 2 * T * 8, base + 3 * T * 8, reset=True)  # stage C (2nd stream)
        self._stage_b(sks, ci, rs["h_hs"].ptr)                                                    # stage B (main)
        self.hip.stage_c_join()
        rs["h_acc"].fetch_async(base)
        rs["ev"].record()
        return dict(slot=slot, sks=sks, shard=shard, ci=ci)

    def finish_pass(self, q, want_multimapped):
        rs = self._result_sets()[q["slot"]]
        T = self.ntax
        rs["ev"].synchronize()  # this pass only: the next one may already be running
        sks, shard = q["sks"], q["shard"]
        for ki, sk in enumerate(sks):
            if sk.resolve():  # stage A's counting table overflowed and the sketch was rebuilt: stage B again
                self._stage_b_again(sks, ki, q["ci"], rs["h_hs"].ptr)
        hs, acc = self._hs_split(rs["h_hs"].array), rs["h_acc"].array.copy()
        mm = shard.multimapped() if want_multimapped else None
        shard.free()
        return sks, hs, (acc[:T], acc[T:2 * T], acc[2 * T:3 * T], acc[3 * T:], mm)

    def profile_commit_launch(self, incoming, first_shard, group_base):
        """Asynchronous part of the commit: accumulator reset + the stage-C pass; nothing is read back."""
        T = self.ntax
        base = self.d_acc.ptr
        self.shard.commit(incoming, first_shard, group_base, base, base + T * 8, base + 2 * T * 8, base + 3 * T * 8,
                          reset=True)

    def profile_commit_finish(self, want_multimapped=True):
        T = self.ntax
        acc = self.d_acc.download()
        mm = self.shard.multimapped() if want_multimapped else None
        self.shard.free()
        return acc[:T], acc[T:2 * T], acc[2 * T:3 * T], acc[3 * T:], mm

    def profile_commit(self, incoming, first_shard, group_base, want_multimapped=True):
        self.profile_commit_launch(incoming, first_shard, group_base)
        return self.profile_commit_finish(want_multimapped)

    # ---- the device side of ShardJob._run_exchange_pipelined (four passes in flight): every method queues work and
    # returns, except the x_wait_* / x_collect ones, which wait for ONE event recorded a tick earlier.
    def x_setup(self, W, G, T, bounds, nslot):
        """bounds: per k, the W+1 hash-range bounds."""
        hip, g, K = self.hip, self.gpad, self.nk
        # (words: per SKETCHED k; the reduce buffer: per column)
        self._xW, self._xNW, self._xnred = W, self.nsk * (W + 4) + 3, 2 * K * G + 2 * T + W * T + K + 2
        if not hasattr(self, "_xs"):
            self._xs = [dict(d_acc=hip.empty(3 * T + 2, np.uint64), h_acc=hip.pinned(3 * T + 2, np.uint64),
                             h_hs=hip.pinned(2 * g * K, np.uint32), h_words=hip.pinned(W * self._xNW, np.int64),
                             h_red=hip.pinned(self._xnred, np.int64), h_red_in=hip.pinned(self._xnred, np.int64),
                             d_red=hip.empty(self._xnred, np.int64), ev=hip.event()) for _ in range(nslot)]
            self._xbounds = [hip.array(np.asarray(b[1:W] if W > 1 else [0], dtype=np.uint64)) for b in bounds]
        if getattr(self, "kmer", False):
            self.reserve_counters(nslot + 4)  # (four passes in flight and three fronts queued ahead of them)

    def x_begin(self):
        # four of the hashing kernel's five workgroups per CU: with no host wait left in the chain the small kernels
        # need fewer issue slots than on the one-pass-at-a-time path (two), but not none (measured on one GPU with
        # every collective in the path: 2 -> 0.735 ms per pass, 3 -> 0.68, 4 -> 0.665, 5 -> 0.71; alternating two
        # stage-A streams here: worse at every setting)
        # (round 3: the fused multi-k kernel is held to 128 VGPRs — four of its workgroups per CU take the WHOLE register
        # file and every other kernel of the tick waits for one of them to retire: three per CU.  configs[3] shapes at world
        # size 1, ms per pass: 4 -> 54.1, 3 -> 52.2, 2 -> 61.1; with the table's resident index 3 -> 39.5, 4 -> 39.1, and the two
        # stage-A streams in turn 39.6 / 39.0: still no better (traced: two hashing kernels side by side take 41 ms each and
        # starve the radix sort of the touched-hash lists, 16 ms).  MG_STAGE_A_WG_PER_CU overrides, for measurements)
        # (... and with the index at half load 3 -> 36.7, 4 -> 35.6, 5 -> 35.7: four for a job that counts in resident indexes)
        resident = any(f is not None and f.resident_bytes for f in self.filters)
        self.hip.stage_a_workgroups_per_cu(int(os.environ.get("MG_STAGE_A_WG_PER_CU", 4 if resident else self.wg_per_cu_exchange)))
        self.hip.stage_a_side_stream(True)

    def x_end(self):
        self.hip.stage_a_side_stream(False)

    def x_front(self, ks, hmaxs, s, pct_id):
        if getattr(self, "kmer", False):
            # the counting kernels of consecutive passes on the two stage-A streams in turn: the next one's launch does not wait
            # for the previous one's last wavefronts (measured for the hash path's sketch, x_begin: worse; for this kernel: better)
            self._xturn = 1 - getattr(self, "_xturn", 1)
            self.hip.stage_a_side_stream(1 + self._xturn)
        return dict(sks=self.sketch_local_async(ks, hmaxs, s), shard=self.new_shard_async(pct_id))

    def x_words(self, P, slot):
        """This rank's words, assembled on the device from the still pending sketches and the map-only pass."""
        P["rs"] = self._xs[slot]
        W, t = self._xW, self.torch
        kmer = getattr(self, "kmer", False)  # (no sketch to cut into slices: those words stay zero)
        word_t = (t.zeros if kmer else t.empty)(self._xNW, dtype=t.int64, device="cuda")
        for ki, sk in enumerate(P["sks"]):
            if not kmer:
                sk.slice_words_dev(self._xbounds[ki].ptr, W - 1, word_t.data_ptr() + 8 * ki * (W + 4))
        P["shard"].map_words_dev(word_t.data_ptr() + 8 * self.nsk * (W + 4))
        return word_t

    def x_redo_words(self, P, bounds, tail):
        """Host path, after a table overflow made the published words stale: settle the sketches and cut them again."""
        W, word = self._xW, []
        for ki, sk in enumerate(P["sks"]):
            sk.resolve()
            n = sk.size
            cuts = [0] + self.split_sketch(sk, bounds[ki][1:W]) + [n]
            last = sk.last_hash
            word += [cuts[q + 1] - cuts[q] for q in range(W)] + [int(sk.truncated), last - (1 << 64) if last >= (1 << 63) else last, n, 0]
        word += list(tail)
        return self.torch.as_tensor(np.asarray(word, dtype=np.int64), device="cuda")

    def x_fetch_words(self, P, words_t, hold):
        rs = P["rs"]
        P["hold"] = (words_t, hold)
        rs["h_words"].fetch_async(words_t.data_ptr())
        rs["ev"].record()

    def x_wait_words(self, P):
        rs = P["rs"]
        rs["ev"].synchronize()
        P["hold"] = None
        return rs["h_words"].array.reshape(self._xW, self._xNW).tolist()

    def x_commit(self, P, incoming, first_shard, group_base):
        T, base = self.ntax, P["rs"]["d_acc"].ptr
        P["shard"].commit(incoming, first_shard, group_base, base, base + T * 8, base + 2 * T * 8, base + 3 * T * 8,
                          reset=True)

    def x_merge(self, P, rh, rc, k, lo, hi, any_trunc, bound):
        P.setdefault("keep", []).append((rh, rc))  # a deferred merge reads its inputs again if it has to be redone
        if getattr(self, "reftable", None) is not None:
            # the reference pipeline's stage B contains a collective (the prefix bitmaps): a merge that had to be redone a
            # phase later could not repeat it on one rank alone — settled here instead (one sketch per pass, not one per k)
            return self.hip.sketch_merge_dev(rh.data_ptr(), rc.data_ptr(), int(rh.numel()), k, lo, hi, 0, any_trunc, bound)
        return self.hip.sketch_merge_dev_async(rh.data_ptr(), rc.data_ptr(), int(rh.numel()), k, lo, hi, 0, any_trunc, bound)

    def x_stage_b(self, P, merged, ci):
        rs = P["rs"]
        P["merged"] = merged
        self._stage_b(merged, ci, rs["h_hs"].ptr)
        self.hip.stage_c_join()
        rs["h_acc"].fetch_async(rs["d_acc"].ptr)
        rs["ev"].record()

    def x_collect(self, P, ci, want_multimapped):
        rs, T = P["rs"], self.ntax
        rs["ev"].synchronize()
        merged = P["merged"]
        for sk in P["sks"]:
            sk.free()  # their buffers were the all-to-all's send buffers: back to the pool only now
        for ki, m in enumerate(merged):
            if m.resolve():
                # (a reference-pipeline merge is settled in the phase that queues it — x_merge: its redo needs the prefix-bitmap collective on
                # every rank — so a rebuilt sketch cannot turn up here; if it ever does, the columns below would be stale)
                if self.reftable is not None:
                    raise RuntimeError("a reference-pipeline sketch was rebuilt after its stage B had run: x_merge must settle it")
                self.hip.sync()
                self.hip.containment_dev(m, self.tables[ki], ci, *self._hs_ptrs(rs["h_hs"].ptr, ki))
                self.hip.sync()
        (hits, sizes), acc = self._hs_split(rs["h_hs"].array), rs["h_acc"].array
        mm = P["shard"].multimapped() if want_multimapped else None
        qn = [m.size for m in merged]
        P["shard"].free()
        for m in merged:
            m.free()
        P["keep"] = None
        return hits, sizes, acc[:T], acc[T:2 * T], acc[2 * T:3 * T], acc[3 * T:], mm, qn

    def x_reduce_buffer(self, P):
        buf = P["rs"]["h_red_in"].array
        buf[:] = 0
        return buf

    def x_reduce_tensor(self, P):
        rs = P["rs"]
        rs["h_red_in"].push_async(rs["d_red"].ptr)
        return self.torch.as_tensor(_CudaView(rs["d_red"].ptr, self._xnred, "<i8"), device="cuda")

    def x_fetch_reduced(self, P, tb):
        rs = P["rs"]
        P["hold"] = tb
        rs["h_red"].fetch_async(rs["d_red"].ptr)
        rs["ev"].record()

    def x_wait_reduced(self, P):
        rs = P["rs"]
        rs["ev"].synchronize()
        P["hold"] = None
        return rs["h_red"].array.copy()


def table_max_hash(dbh, dbo):
    """Largest hash of a genome-major table (every genome sketch is ascending: its maximum is its last entry)."""
    dbo = np.asarray(dbo)
    tails = dbo[1:][dbo[1:] > dbo[:-1]] - 1
    return int(np.asarray(dbh)[tails.astype(np.int64)].max()) if len(tails) else 0


class _HostStagedGloo:
    """torch.distributed over GLOO for tensors that live on the GPU, staged through host memory (gloo moves CPU tensors).
    RCCL is the product's transport; this exists so that the multi-rank path can run with the REAL kernels where RCCL
    cannot — two ranks on one GPU ("Duplicate GPU detected") — i.e. for tests on single-GPU machines."""

    def __init__(self, dist, torch):
        self._d, self._t = dist, torch

    def __getattr__(self, name):  # get_backend, barrier, ReduceOp, get_rank, ...
        return getattr(self._d, name)

    def all_gather(self, outs, t):
        cin = t.cpu()
        couts = [self._t.empty(o.shape, dtype=o.dtype) for o in outs]
        self._d.all_gather(couts, cin)
        for o, c in zip(outs, couts):
            o.copy_(c)

    def all_reduce(self, t, op=None):
        c = t.cpu()
        self._d.all_reduce(c, op=op if op is not None else self._d.ReduceOp.SUM)
        t.copy_(c)

    class _Op:
        def __init__(self, fn, tensor, peer):
            self.fn, self.tensor, self.peer = fn, tensor, peer

    def P2POp(self, fn, tensor, peer):
        return self._Op(fn, tensor, peer)

    def batch_isend_irecv(self, ops):
        staged = []
        for op in ops:
            c = op.tensor.cpu() if op.fn is self._d.isend else self._t.empty(op.tensor.shape, dtype=op.tensor.dtype)
            staged.append((op, c))
        reqs = self._d.batch_isend_irecv([self._d.P2POp(op.fn, c, op.peer) for op, c in staged])
        for r in reqs:
            r.wait()
        for op, c in staged:
            if op.fn is self._d.irecv:
                op.tensor.copy_(c)
        return []


def selfcheck_collectives(dist, torch, rank, world, device):
    """Known-answer run of every collective a pass uses, in the shapes it uses them, before any real work: an
    all-gather of a few int64 words, an all-to-all with UNEVEN splits (rank r sends q + 1 + (r + q) % 3 entries to
    rank q; gloo: the same movement as point-to-point operations, exactly as ShardJob._all_to_all does), an
    all-reduce(sum, int64).  Raises RuntimeError naming the collective whose result is wrong — a transport that
    delivers wrong or misplaced data must stop the job at start-up, not show up as a wrong containment index.
    (scripts/select_db.py:73-76 and scripts/map_and_profile.py:193-264 are single-process; the collectives are this
    build's own and so is their check.)"""
    d = _HostStagedGloo(dist, torch) if (str(device) != "cpu" and dist.get_backend() == "gloo") else dist
    t, W = torch, world
    # all-gather
    mine = t.as_tensor([rank, 1000 + rank, -(rank + 1)], dtype=t.int64, device=device)
    got = [t.zeros(3, dtype=t.int64, device=device) for _ in range(W)]
    d.all_gather(got, mine)
    for q in range(W):
        if got[q].cpu().tolist() != [q, 1000 + q, -(q + 1)]:
            raise RuntimeError("collective self-check: all_gather delivered %r for rank %d on rank %d" % (got[q].cpu().tolist(), q, rank))
    # all-to-all, uneven splits: entry j of the slice r -> q carries r * 10^6 + q * 10^3 + j
    n = lambda r, q: q + 1 + (r + q) % 3  # noqa: E731
    send_counts = [n(rank, q) for q in range(W)]
    recv_counts = [n(q, rank) for q in range(W)]
    send = t.as_tensor(np.concatenate([np.arange(n(rank, q)) + rank * 10**6 + q * 10**3 for q in range(W)]).astype(np.int64), device=device)
    recv = t.zeros(sum(recv_counts), dtype=t.int64, device=device)
    if d.get_backend() == "nccl":
        d.all_to_all_single(recv, send, list(recv_counts), list(send_counts))
    else:
        so, ro = np.cumsum([0] + send_counts), np.cumsum([0] + recv_counts)
        ops = []
        for q in range(W):
            if q == rank:
                recv[ro[q]:ro[q + 1]] = send[so[q]:so[q + 1]]
                continue
            ops.append(d.P2POp(d.isend, send[so[q]:so[q + 1]].contiguous(), q))
            ops.append(d.P2POp(d.irecv, recv[ro[q]:ro[q + 1]], q))
        if ops:
            for req in d.batch_isend_irecv(ops):
                req.wait()
    want = np.concatenate([np.arange(n(q, rank)) + q * 10**6 + rank * 10**3 for q in range(W)])
    if not np.array_equal(recv.cpu().numpy(), want):
        raise RuntimeError("collective self-check: all-to-all with uneven splits delivered wrong data on rank %d" % rank)
    # all-reduce(sum, int64), values above 2^32 so that a 32-bit reduction would show
    v = t.as_tensor([(rank + 1) * (1 << 33), rank, 1], dtype=t.int64, device=device)
    d.all_reduce(v, op=d.ReduceOp.SUM)
    if v.cpu().tolist() != [W * (W + 1) // 2 * (1 << 33), W * (W - 1) // 2, W]:
        raise RuntimeError("collective self-check: all_reduce(sum, int64) gave %r on rank %d" % (v.cpu().tolist(), rank))


class ShardJob:
    """One rank's share of a sample and the collective choreography around it."""

    def __init__(self, hip, dist, rank, world, k, ci=2, pct_id=0.5, s=0, engine=None, always_exchange=False,
                 definition="sketch_per_k", match=None):
        """k: one k-mer size or a sequence of them (ascending; the reference's cutoff reads the largest).
        definition: "sketch_per_k" — every k has a genome table of its own and the reads are sketched at every k; or
        "reference_pipeline" — stage A/B wired as scripts/select_db.py:50-59,73-76 wires KMC and CMash: the reads are sketched
        at the LARGEST k only and every smaller k's column comes from the k-prefixes of the matched k_max-mers (load() then
        takes the reference pipeline's table: include/metalign_hip.h, mg_refdb).
        match (the reference pipeline): how a read k_max-mer meets a sketched one — "kmer": by what it IS, as `kmc` +
        `kmc_tools intersect` do (scripts/select_db.py:50-59; mg_kcount.hip: no hash on the read side; 15 <= k_max <= 64; every
        rank of a multi-GPU job holds the whole table and counts ITS reads, the ranks' counters — two bits per pair at the
        reference's -cs3 — are all-gathered and summed: _sum_kmer_counts); "hash": by its MurmurHash3 value (rounds 4-5; any k;
        the table sharded by hash range); None: "kmer" for k_max from 25 to 64 when the table holds its k-mers, "hash" otherwise."""
        self.dist, self.rank, self.world = dist, rank, world
        # always_exchange: run the collectives even when world == 1 (single-GPU validation of the RCCL path)
        self.exchange = dist is not None and (world > 1 or always_exchange)
        self.single_k = np.isscalar(k)
        self.ks = [int(k)] if self.single_k else [int(x) for x in k]
        self.k = self.ks[-1]
        self.definition = definition
        self.refpipe = definition == "reference_pipeline"
        if definition not in ("sketch_per_k", "reference_pipeline"):
            raise ValueError("unknown stage A/B definition %r" % (definition,))
        if self.refpipe and s:
            raise ValueError("the reference pipeline counts every k_max-mer of the reads: no bottom-s sketch (s = %d)" % s)
        self.sks_k = [self.ks[-1]] if self.refpipe else self.ks  # the k the READS are sketched at
        if match not in (None, "kmer", "hash"):
            raise ValueError("match is 'kmer' or 'hash', not %r" % (match,))
        # (an engine of the tests serves it when it says so: tests/test_distributed_gloo.py, the exchange on the CPU)
        can_kmer = self.refpipe and KMER_MATCH_MIN_K <= self.ks[-1] <= KMER_MATCH_MAX_K and (engine is None or getattr(engine, "supports_kmer", False))
        if match == "kmer" and not can_kmer:
            raise ValueError("match='kmer' needs the reference pipeline, the library's engine and 15 <= k_max <= 64")
        by_default = can_kmer and kmer_match_by_default(self.ks[-1])  # (a small k_max: the read sketch is the faster of the two)
        self.match = "kmer" if (can_kmer and (match == "kmer" or (match is None and by_default))) else ("hash" if self.refpipe else None)
        self._match_asked = match  # (None: a table that does not hold its k-mers falls back to "hash" at load())
        self.ci, self.pct_id, self.s = ci, pct_id, s
        if engine is not None:
            self.engine = engine
            self.torch = engine.torch
        else:
            tm = None
            if dist is not None:
                import torch as tm  # noqa: F811
            self.torch = tm
            if tm is not None and self.exchange:
                # torch / RCCL collectives order against torch's CURRENT stream only: it must be the stream the library
                # launches on (mg_init_on_stream), or the all-gather / all-to-all / all-reduce race with its kernels
                cur = tm.cuda.current_stream().cuda_stream
                if getattr(hip, "main_stream", None) != cur:
                    raise _hip.HipError("ShardJob: the library's main stream (%r) is not torch's current stream (%r); "
                                        "create a torch.cuda.Stream, make it current and pass it to Hip.get(device, "
                                        "stream=...) before anything else initialises the library"
                                        % (getattr(hip, "main_stream", None), cur))
            self.engine = HipEngine(hip, tm)
        self.device = getattr(self.engine, "device", "cuda")
        if dist is not None and self.device != "cpu" and dist.get_backend() == "gloo":
            self.dist = _HostStagedGloo(dist, self.torch)  # (tests: several ranks on one GPU)

    def load(self, rbases, roffsets, recs, ref2tax, dbh, dbo=None, ntax=None, reftable=None):
        """recs: this rank's shard (starts on a read boundary).
        The reference pipeline (definition="reference_pipeline"): dbh = the table's host arrays — a formats.SketchTable of
        version 3, or the dict _hip.RefTable.download() returns — of which this rank uploads its share (pairs by hash range,
        count lists by prefix range); reftable = the whole table already on the device (world size 1: nothing is uploaded).
        Tables, one per k (a single k takes them bare): either dbh / dbo = the FULL genome-major table (hashes,
        offsets[G+1]), sliced here by hash range; or dbh = a formats.SketchTable whose files are hash-major — then only
        this rank's hash range [bounds[r], bounds[r+1]) of every k is read from disk (dbo stays None).
        ntax: number of dense taxon ids (default: max(ref2tax) + 1)."""
        K = len(self.sks_k)
        if hasattr(self.engine, "wants_resident_index"):
            self.engine.nsk_hint = K  # (its memory estimate covers every sketched k's index)
            self.engine.bottom_s = self.s  # (a bottom-s sketch keeps the bit filter's definition: no index then)
        self.T = int(ntax) if ntax is not None else (int(np.max(ref2tax)) + 1 if len(ref2tax) else 0)
        tables, self.hmaxs, self.bounds = [], [], []
        if self.refpipe:
            reftable = self._load_refpipe(dbh, reftable)
        elif hasattr(dbh, "pairs"):  # an on-disk hash-major table (formats.SketchTable, version 2)
            disk = dbh
            self.G = disk.ngenomes
            for ki, k in enumerate(self.ks):
                hmax = disk.max_hash(k)
                b = table_bounds(disk.pairs(k)["pair_hash"], self.world, hmax, presorted=True)
                self.hmaxs.append(hmax)
                self.bounds.append(b)
                if hasattr(self.engine, "set_filter"):
                    bits = disk.filter_bits(k)  # stored by the builder: the rank does not stream the other ranks' slices
                    if hasattr(self.engine, "wants_resident_index") and \
                            self.engine.wants_resident_index(hmax, len(disk.pairs(k)["pair_hash"])):
                        bits = None  # the resident index wants every hash of the table on every rank
                    if bits is not None and hasattr(self.engine, "set_filter_bits"):
                        self.engine.set_filter_bits(ki, bits)
                    else:
                        self.engine.set_filter(ki, disk.pairs(k)["pair_hash"])
                tables.append(disk.pairs(k, b[self.rank], b[self.rank + 1]))
        else:
            per_k = [(dbh, dbo)] if self.single_k else list(zip(dbh, dbo))
            assert len(per_k) == K, "one (hashes, offsets) table per k"
            self.G = len(per_k[0][1]) - 1
            for ki, (h, o) in enumerate(per_k):
                hmax = table_max_hash(h, o)
                b = table_bounds(h, self.world, hmax)
                self.hmaxs.append(hmax)
                self.bounds.append(b)
                if hasattr(self.engine, "set_filter"):
                    self.engine.set_filter(ki, h)  # from the FULL table, before it is sliced
                if self.world > 1:
                    h, o = table_slice(h, o, b[self.rank], b[self.rank + 1])
                tables.append((h, o))
        self.hmax = self.hmaxs[-1]
        has_look = False
        if self.exchange:
            # the first record of the NEXT non-empty shard closes this shard's last read (:225-226)
            t = self.torch
            mine = np.zeros(5, dtype=np.int64)
            if len(recs):
                mine[0] = 1
                mine[1:] = [int(recs[0][f]) for f in ("ref_new", "matched", "total", "flag_len")]
            gathered = [t.zeros(5, dtype=t.int64, device=self.device) for _ in range(self.world)]
            self.dist.all_gather(gathered, t.as_tensor(mine, device=self.device))
            heads = [g.cpu().numpy() for g in gathered]
            nxt = next((h for h in heads[self.rank + 1:] if h[0]), None)
            self.nonempty = [bool(h[0]) for h in heads]
            if nxt is not None and len(recs):
                look = np.zeros(1, dtype=_hip.REC_DTYPE)
                look[0] = tuple(int(v) for v in nxt[1:])
                recs = np.concatenate([recs, look])
                has_look = True
        else:
            self.nonempty = [len(recs) > 0]
        if self.refpipe:
            self.engine.mark_exchange = self._or_marks if (self.exchange and self.match != "kmer") else None
            self.engine.match_kmer = self.match == "kmer"
            self.engine.count_share = (self.rank, self.world) if (self.exchange and self.match == "kmer") else None
            self.engine.load(rbases, roffsets, recs, has_look, ref2tax, self.T, [], reftable=reftable)
        else:
            self.engine.load(rbases, roffsets, recs, has_look, ref2tax, self.T, tables)
        if hasattr(self.engine, "prime") and (len(roffsets) > 1 or self.exchange):
            choice = self.engine.prime(self.sks_k, self.hmaxs, self.s) if len(roffsets) > 1 else None
            # (all(): a rank on which only SOME k got an index — no room, crowded buckets — does not count as having one:
            # every rank, and every k of a rank, ends up on the same side)
            fl = [f for f in getattr(self.engine, "filters", [])[: len(self.sks_k)]]
            resident = bool(fl) and all(f is not None and f.resident_bytes for f in fl)
            some = any(f is not None and f.resident_bytes for f in fl)
            drop = (some and not resident) or (resident and choice is not None and choice[1] < choice[0])
            if self.exchange and self.world > 1:
                # every rank takes the same side (the ranks' sketches are slices of ONE sketch): an index everywhere or
                # nowhere (a rank may have had no room for it), and the SUM of the ranks' timings decides
                t = self.torch
                v = t.as_tensor([1.0 if resident else 0.0] + (list(choice) if choice else [0.0, 0.0]), dtype=t.float64,
                                device=self.device)
                self.dist.all_reduce(v)
                have, with_index, with_filter = (float(x) for x in v.cpu())
                drop = have < self.world or with_filter < with_index
                choice = (with_index, with_filter) if have else None
            if choice is not None:
                self.resident_choice = dict(with_index_s=choice[0], with_filter_s=choice[1])
            if drop and some:
                self.engine.drop_resident_indexes()
            # the priming pass was sized for the worst case (no distinct-count ratio yet: tens of GB against a dense
            # table); its blocks would stay cached for the life of the process
            if hasattr(self.engine, "hip"):
                self.engine.hip.mem_trim()
        if hasattr(self.engine, "hip"):
            # stage C runs on the library's second stream: its latency-bound passes overlap stage A's tail, stage B
            # and (with the exchange) the collectives  (MG_SINGLE_STREAM=1: everything on one stream, for profiles in
            # which no two kernels overlap)
            self.engine.hip.stage_c_side_stream(os.environ.get("MG_SINGLE_STREAM", "0") != "1")

    def _load_refpipe(self, src, reftable):
        """The reference pipeline's table for this rank: hash-range bounds and the pre-filter from ALL pairs of the largest k;
        the pairs of this rank's hash range with their prefix numbers, and of every smaller k's count list the run whose prefix
        numbers fall in this rank's share of [0, nprefix) (cut on multiples of 32: a rank's share of a prefix bitmap is whole
        words).  -> the engine's table handle."""
        eng, W = self.engine, self.world
        disk = hasattr(src, "refpipe_arrays")
        full = src.refpipe_arrays() if disk else src
        if [int(k) for k in full["ks"]] != self.ks:
            raise ValueError("the table holds k = %r, the job was made for %r" % (list(full["ks"]), self.ks))
        self.G = int(full["ngenomes"])
        ph = full["pair_hash"]
        hmax = int(full["max_hash"]) if "max_hash" in full else (int(ph[-1]) if len(ph) else 0)
        b = table_bounds(ph, W, hmax, presorted=True)
        self.hmaxs, self.bounds = [hmax], [b]
        small_all = full["small"] if isinstance(full["small"], list) else [full["small"][k] for k in self.ks[:-1]]
        self.nprefix = [int(t["nprefix"]) for t in small_all]
        # word-aligned prefix cuts per smaller k: rank r counts prefixes [32 * cut[r], 32 * cut[r + 1])
        self.mark_cuts = [[((npre + 31) // 32) * r // W for r in range(W)] + [(npre + 31) // 32] for npre in self.nprefix]
        if self.match == "kmer":
            # no filter, no resident index: the read side hashes nothing.  The index over the table's k-mers (built once, on the
            # device) from the k-mers a table built here holds, or from the stored ones
            if reftable is None and full.get("kmer_hi") is None and self._match_asked is None:
                self.match = "hash"  # (nobody asked for it, and this table cannot serve it)
        if self.match == "kmer":
            if reftable is None:
                if full.get("kmer_hi") is None:
                    raise ValueError("match='kmer' needs the table's k-mers (format 3: k<K>.kmer_hi.u64 / .kmer_lo.u64); "
                                     "this table has none: ShardJob(match='hash')")
                if not hasattr(eng, "hip"):  # (a host engine of the tests takes the arrays themselves: the WHOLE table)
                    return dict(full, ks=self.ks, ngenomes=self.G)
                reftable = eng.hip.refdb_upload(self.ks, self.G, ph, full["pair_gen"], full["gsize"], hmax, small_all)
                reftable.index_kmers(full["kmer_hi"], full["kmer_lo"])
            elif not reftable.has_kmer_index:
                reftable.index_kmers()
            return reftable
        if hasattr(eng, "set_filter"):
            bits = src.filter_bits(self.ks[-1]) if disk else None
            if hasattr(eng, "wants_resident_index") and eng.wants_resident_index(hmax, len(ph)):
                bits = None
            if bits is not None and hasattr(eng, "set_filter_bits"):
                eng.set_filter_bits(0, bits)
            else:
                eng.set_filter(0, ph)
        if reftable is not None and W == 1:
            return reftable
        if W == 1:
            share = dict(pair_hash=ph, pair_gen=full["pair_gen"], gsize=full["gsize"], small=small_all)
        else:
            r = self.rank
            a = int(np.searchsorted(ph, np.uint64(b[r]), side="left"))
            z = len(ph) if b[r + 1] > U64_MAX else int(np.searchsorted(ph, np.uint64(b[r + 1]), side="left"))
            pg = np.asarray(full["pair_gen"][a:z])
            small = []
            for ki, t in enumerate(small_all):
                lo, hi = 32 * self.mark_cuts[ki][r], 32 * self.mark_cuts[ki][r + 1]
                ca, cb = int(np.searchsorted(t["cid"], lo, side="left")), int(np.searchsorted(t["cid"], hi, side="left"))
                cg = np.asarray(t["cgen"][ca:cb])
                small.append(dict(pa=t["pa"][a:z], pb=t["pb"][a:z], cid=t["cid"][ca:cb], cgen=cg,
                                  gsize=np.bincount(cg, minlength=self.G).astype(np.uint32), nprefix=t["nprefix"]))
            share = dict(pair_hash=ph[a:z], pair_gen=pg, gsize=np.bincount(pg, minlength=self.G).astype(np.uint32), small=small)
        if hasattr(eng, "hip"):
            return eng.hip.refdb_upload(self.ks, self.G, share["pair_hash"], share["pair_gen"], share["gsize"], hmax, share["small"])
        share.update(ks=self.ks, ngenomes=self.G)
        return share  # (a host engine of the tests takes the arrays themselves)

    def _sent(self, kind, nbytes, entries=0):
        """Bytes this rank hands to OTHER ranks, by collective, summed over the passes so far (traffic["passes"]); what stays
        on the rank — its own slice of an all-to-all — is not counted.  `sketch_entries` counts every entry it routes, its
        own slice included: at world size 1 that is the volume that would be split W ways."""
        tr = self.__dict__.setdefault("traffic", {"passes": 0, "words_all_gather": 0, "sketch_all_to_all": 0, "sketch_entries": 0,
                                                  "prefix_marks_all_to_all": 0, "kmer_counts_all_gather": 0, "results_all_reduce": 0})
        tr[kind] += int(nbytes)
        tr["sketch_entries"] += int(entries)

    def traffic_per_pass(self):
        """-> bytes per pass this rank sent, by collective (all_reduce: the payload; a ring moves 2 (W - 1) / W of it per rank)."""
        tr = getattr(self, "traffic", None)
        if not tr or not tr["passes"]:
            return None
        n = tr["passes"]
        out = {k: v / n for k, v in tr.items() if k != "passes"}
        out["total_bytes"] = sum(v for k, v in out.items() if k != "sketch_entries")
        out["passes"] = n
        return out

    def _or_marks(self, marks):
        """The reference pipeline's one extra exchange.  marks: per k below the largest, this rank's bitmap over ALL prefixes of
        the table (int32 words; the prefixes of the k_max-mers that matched in this rank's hash range).  -> per k a bitmap of the
        same size in which the words of THIS rank's prefix share hold the OR over the ranks (what its runs of the count lists
        read): an all-to-all of word ranges — bytes per rank do not grow with the world size — and W - 1 ORs."""
        t, dist, W, r = self.torch, self.dist, self.world, self.rank
        out = []
        for ki, m in enumerate(marks):
            cuts = self.mark_cuts[ki]
            if W == 1 or m.numel() == 0:
                out.append(m)
                continue
            sizes = [cuts[q + 1] - cuts[q] for q in range(W)]
            my = sizes[r]
            self._sent("prefix_marks_all_to_all", (int(m.numel()) - my) * m.element_size())
            if dist.get_backend() == "nccl":
                recv = t.empty(W * my, dtype=m.dtype, device=m.device)
                dist.all_to_all_single(recv, m, [my] * W, sizes)
                red = recv[:my].clone()
                for q in range(1, W):
                    red |= recv[q * my:(q + 1) * my]
            else:  # gloo (tests): no all-to-all — the whole bitmaps gathered
                parts = [t.empty_like(m) for _ in range(W)]
                dist.all_gather(parts, m.contiguous())
                full = parts[0].clone()
                for q in range(1, W):
                    full |= parts[q]
                red = full[cuts[r]:cuts[r + 1]]
            mine = t.zeros_like(m)
            mine[cuts[r]:cuts[r + 1]] = red
            out.append(mine)
        return out

    def _sum_kmer_counts(self, kc):
        """Stage A by k-mer identity on several ranks: this rank's counters (of ITS reads against the whole table) become the
        sample's.  Counters that saturate at 3 or below (kmc -cs3, the reference's setting): min(counter, 3) of every pair in
        two bits, ONE all-gather of those arrays (2.5 MB per ten million pairs and rank) and their sum on every rank;
        otherwise the 32-bit counters themselves through an all-reduce.  Everything is queued on the main stream (torch's
        current one); -> the tensors that must outlive the queue."""
        t, dist, W, eng = self.torch, self.dist, self.world, self.engine
        cs = eng.kmer_saturation()
        if 1 <= cs <= 3:
            mine = eng.kmer_pack(kc)
            every = t.empty((W, int(mine.numel())), dtype=t.int32, device=mine.device)
            dist.all_gather(list(every.unbind(0)), mine)
            self._sent("kmer_counts_all_gather", 4 * int(mine.numel()) * (W - 1))
            eng.kmer_merge(kc, every)
            return mine, every
        tv = eng.kmer_raw(kc)
        dist.all_reduce(tv, op=dist.ReduceOp.SUM)
        self._sent("kmer_counts_all_gather", 4 * int(tv.numel()) if W > 1 else 0)
        return (tv,)

    # ------------------------------------------------------------------
    def _all_to_all(self, send_h, send_c, send_counts, recv_counts):
        """Variable all-to-all of the sketch arrays: send_counts[q] consecutive entries go to rank q; the slices
        are contiguous ranges of the ascending sketch, so the sketch's own buffers are the send buffers."""
        t, dist, W = self.torch, self.dist, self.world
        rh = t.zeros(sum(recv_counts), dtype=t.int64, device=self.device)
        rc = t.zeros(sum(recv_counts), dtype=t.int32, device=self.device)
        self._sent("sketch_all_to_all", 12 * (sum(send_counts) - send_counts[self.rank]), entries=sum(send_counts))
        if dist.get_backend() == "nccl":
            w1 = dist.all_to_all_single(rh, send_h, list(recv_counts), list(send_counts), async_op=True)
            w2 = dist.all_to_all_single(rc, send_c, list(recv_counts), list(send_counts), async_op=True)
            return rh, rc, [w1, w2]  # in flight on RCCL's stream: the caller overlaps stage C's commit with it
        # gloo has no all-to-all: point-to-point exchange with the same data movement
        so, ro = np.cumsum([0] + list(send_counts)), np.cumsum([0] + list(recv_counts))
        ops = []
        for q in range(W):
            if q == self.rank:
                rh[ro[q]:ro[q + 1]] = send_h[so[q]:so[q + 1]]
                rc[ro[q]:ro[q + 1]] = send_c[so[q]:so[q + 1]]
                continue
            if send_counts[q]:
                ops.append(dist.P2POp(dist.isend, send_h[so[q]:so[q + 1]].contiguous(), q))
                ops.append(dist.P2POp(dist.isend, send_c[so[q]:so[q + 1]].contiguous(), q))
            if recv_counts[q]:
                ops.append(dist.P2POp(dist.irecv, rh[ro[q]:ro[q + 1]], q))
                ops.append(dist.P2POp(dist.irecv, rc[ro[q]:ro[q + 1]], q))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return rh, rc, []

    def _merge_args(self, words, ki):
        """From the gathered words of k number ki: (any source truncated?, the completeness bound, this rank's range)."""
        W, o = self.world, ki * (self.world + 4)
        # completeness: a source truncated at its s-th hash knows nothing above it
        lasts = [_u64(w[o + W + 1]) for w in words if w[o + W] and w[o + W + 2]]
        complete_to = min(lasts) if lasts else U64_MAX
        b = self.bounds[ki]
        return bool(lasts), complete_to, b[self.rank], b[self.rank + 1] - 1

    def _exchange_step(self):
        """Stage A + pass A of stage C locally, then ONE all-gather of per-rank words (slice sizes and sketch
        completeness of every k, carried-state map) and ONE all-to-all round of sketch slices, during which stage C's
        commit runs (it only needs the gathered state maps).
        -> (this rank's slice of the sample sketch for every k, commit results)."""
        eng, t, dist, W, K = self.engine, self.torch, self.dist, self.world, len(self.sks_k)
        if hasattr(eng, "profile_begin_async"):
            # stage C's map-only pass is queued first: stage A's one synchronisation covers it too
            if self._given is not None:
                sks, eng.shard = self._given  # queued a pass ahead by run()
            else:
                eng.profile_begin_async(self.pct_id)
                sks = eng.sketch_local(self.sks_k, self.hmaxs, self.s)
            (m0, m1), ngroups = eng.profile_map()
        else:
            sks = eng.sketch_local(self.sks_k, self.hmaxs, self.s)
            (m0, m1), ngroups = eng.profile_begin(self.pct_id, True)
        kmer = self.match == "kmer"
        word, send_counts = [], []
        if kmer:
            word = [0] * (K * (W + 4))
        for ki, sk in enumerate([] if kmer else sks):
            n = sk.size
            cuts = [0] + eng.split_sketch(sk, self.bounds[ki][1:W]) + [n]
            sc = [cuts[q + 1] - cuts[q] for q in range(W)]
            send_counts.append(sc)
            last = sk.last_hash  # two's complement into the int64 word
            word += sc + [int(sk.truncated), last - (1 << 64) if last >= (1 << 63) else last, n, 0]
        word += [m0, m1, ngroups]
        NW = K * (W + 4) + 3
        words = [t.zeros(NW, dtype=t.int64, device=self.device) for _ in range(W)]
        dist.all_gather(words, t.as_tensor(np.asarray(word, dtype=np.int64), device=self.device))
        self._sent("words_all_gather", 8 * len(word) * (W - 1))
        words = t.stack(words).cpu().numpy().tolist()
        received, inflight = [], []
        for ki, sk in enumerate([] if kmer else sks):
            recv_counts = [words[p][ki * (W + 4) + self.rank] for p in range(W)]
            h, c = eng.export_sketch(sk)
            rh, rc, fl = self._all_to_all(h, c, send_counts[ki], recv_counts)
            received.append((rh, rc))
            inflight += fl
        # stage C commit overlaps the all-to-all: it depends on the gathered maps only
        tail = K * (W + 4)
        maps = [(w[tail], w[tail + 1]) for w in words]
        incoming = compose_incoming(maps, self.rank)
        group_base = int(sum(w[tail + 2] for w in words[: self.rank]))
        first_shard = self.nonempty[self.rank] and not any(self.nonempty[: self.rank])
        if hasattr(eng, "profile_commit_launch"):
            eng.profile_commit_launch(incoming, first_shard, group_base)  # read back with stage B's counts (step())
            committed = None
        else:
            committed = eng.profile_commit(incoming, first_shard, group_base, self._want_mm)
        if kmer:
            self._kmer_keep = self._sum_kmer_counts(sks[0].counts)  # (until the next pass: stage B is queued behind it)
            return sks, committed
        for wk in inflight:
            wk.wait()
        merged = []
        for ki, (rh, rc) in enumerate(received):
            any_trunc, complete_to, lo, hi = self._merge_args(words, ki)
            m = eng.merge_sketches(rh, rc, self.sks_k[ki], 0, any_trunc, complete_to, (lo, hi))
            if self.s or any_trunc:
                m = self._bottom_s(m, any_trunc, self.sks_k[ki])
            merged.append(m)
        # the sketches' buffers were the all-to-all's send buffers: they go back to the pool only now that the merges
        # (which read what the all-to-all delivered, and synchronised) are done
        for sk in sks:
            sk.free()
        if hasattr(eng, "keep"):
            eng.keep = []
        return merged, committed

    def _bottom_s(self, merged, any_trunc, k):
        """bottom-s over the rank-ordered slices: keep the first s entries of the global order and tell every
        slice the sample's completeness bound (two tiny all-gathers; only when s > 0)."""
        eng, t, dist, W = self.engine, self.torch, self.dist, self.world
        sizes = [t.zeros(1, dtype=t.int64, device=self.device) for _ in range(W)]
        dist.all_gather(sizes, t.as_tensor([merged.size], dtype=t.int64, device=self.device))
        sizes = [int(x[0].item()) for x in sizes]
        before, total = sum(sizes[: self.rank]), sum(sizes)
        keep, truncated = merged.size, any_trunc
        if self.s and total > self.s:
            keep = max(0, min(merged.size, self.s - before))
            truncated = True
        if keep < merged.size:
            mh, mc = eng.export_sketch(merged)
            cut = eng.merge_sketches(mh[:keep].contiguous(), mc[:keep].contiguous(), k, 0, False, 0, None)
            merged.free()
            merged = cut
        mine = 0
        if merged.size:
            mh, _ = eng.export_sketch(merged)
            mine = int(mh[-1].item())
        lasts_t = [t.zeros(2, dtype=t.int64, device=self.device) for _ in range(W)]
        dist.all_gather(lasts_t, t.as_tensor([merged.size, mine], dtype=t.int64, device=self.device))
        kept = [(int(x[0].item()), _u64(x[1].item())) for x in lasts_t]
        sample_last = max((hh for nn, hh in kept if nn), default=0)
        eng.set_sketch_bound(merged, truncated, sample_last)
        return merged

    def run(self, nsteps, want_multimapped=False):
        """`nsteps` passes over the resident batch with several passes in flight (see the two schedules below).  Every
        pass is complete when this returns; the last pass's results are returned."""
        eng = self.engine
        if nsteps < 1:
            return None
        if self.exchange and hasattr(eng, "x_front") and os.environ.get("MG_EXCHANGE_PIPELINE", "1") != "0":
            return self._run_exchange_pipelined(nsteps, want_multimapped)
        if not hasattr(eng, "sketch_local_async"):
            out = None
            for _ in range(nsteps):
                out = self.step(want_multimapped)
            return out
        if not self.exchange:
            # Single shard: no exchange to hide.  The GPU is not left waiting for the host between passes: pass i+1 is
            # queued (behind pass i) BEFORE pass i is read back ...
            # ... and stage A of consecutive passes goes to two alternating streams at full occupancy: pass i+1's
            # k_sketch_reads fills the GPU while pass i's sort / pack / stage B tail (small kernels) drains.
            import gc
            side = os.environ.get("MG_SINGLE_STREAM", "0") != "1"  # (=1: everything on one stream, for clean profiles)
            eng.hip.stage_a_workgroups_per_cu(0)
            gc_was_on = gc.isenabled()
            gc.disable()  # (a cyclic collection in the loop is a hole of milliseconds in the GPU's queue)
            try:
                q = eng.queue_pass(0, self.sks_k, self.hmaxs, self.s, self.ci, self.pct_id, side)
                out = None
                for i in range(nsteps):
                    nxt = (eng.queue_pass((i + 1) & 1, self.sks_k, self.hmaxs, self.s, self.ci, self.pct_id, side)
                           if i + 1 < nsteps else None)
                    sks, (hits, sizes), committed = eng.finish_pass(q, want_multimapped)
                    out = self._results(sks, hits, sizes, committed)
                    q = nxt
            finally:
                if side:
                    eng.hip.stage_a_side_stream(0)
                if gc_was_on:
                    gc.enable()
            return out
        # (MG_EXCHANGE_PIPELINE=0: the older schedule, one pass at a time with stage A a pass ahead)
        eng.hip.stage_a_workgroups_per_cu(2)
        eng.hip.stage_a_side_stream(True)
        try:
            def front():  # what does not depend on the other ranks: stage A and stage C's map-only pass
                return eng.sketch_local_async(self.sks_k, self.hmaxs, self.s), eng.new_shard_async(self.pct_id)
            nxt = front()
            out = None
            for i in range(nsteps):
                cur = nxt
                nxt = front() if i + 1 < nsteps else None
                out = self.step(want_multimapped, _sketch=cur)
            return out
        finally:
            eng.hip.stage_a_side_stream(False)

    # ---- the exchange path with several passes in flight -------------------------------------------------------
    # A pass with an exchange is a chain: local sketch -> [sync] split -> all-gather -> [sync] all-to-all -> merge
    # [sync] -> stage B + stage-C commit -> [sync] all-reduce -> [sync] results.  Run one pass at a time and every
    # one of those waits — each a collective's latency on a real node — is dead time on the host, and the pass rate
    # is 1 / (length of the chain) however fast the kernels are.  Passes are independent of each other, so the chain
    # is cut into four phases that each START by waiting for what the previous phase of the same pass queued and END
    # by queueing asynchronous work; a tick runs phase D of pass t-3, C of t-2, B of t-1, A of t (the same order on
    # every rank, so the collectives match up), stage A of passes t+1 and t+2 is already queued on the stage-A
    # stream.  By the time a phase looks at its inputs a whole tick has passed: the waits find finished work.
    def _red_layout(self):
        G, T, W, K = self.G, self.T, self.world, len(self.ks)
        o_sizes, o_count = K * G, 2 * K * G
        o_bases, o_first = o_count + T, o_count + 2 * T
        o_qn = o_first + W * T
        return o_sizes, o_count, o_bases, o_first, o_qn, o_qn + K, o_qn + K + 2  # ..., scalars, total

    def _fill_reduce(self, buf, hits, sizes, count, bases, first, scalars, qn):
        G, T, K = self.G, self.T, len(self.ks)
        o_sizes, o_count, o_bases, o_first, o_qn, o_scal, _ = self._red_layout()
        if getattr(self, "match", None) == "kmer":
            # every rank holds the whole table and the sample's summed counters: every rank's columns ARE the sample's — rank 0's
            # go into the sum (the matched pairs of the largest k stand in for a sketch size)
            qn = [int(np.asarray(hits)[-1].sum()) if q is None else q for q in qn]
            if self.rank != 0:
                part = np.asarray(hits).copy()
                if getattr(self.engine, "kmer_count_sharded", False):
                    part[-1] = 0  # (the columns of k < k_max are this rank's share of the count lists: they add up; the k_max column is whole)
                else:
                    part[:] = 0
                hits, sizes, qn = part, np.zeros_like(np.asarray(sizes)), [0] * len(qn)
        buf[:K * G] = np.asarray(hits).reshape(-1)  # per-slice partial sums: the all-reduce adds them up
        buf[o_sizes:o_sizes + K * G] = np.asarray(sizes).reshape(-1)
        buf[o_count:o_count + T] = count.view(np.int64)
        buf[o_bases:o_bases + T] = bases.view(np.int64)
        o = o_first + self.rank * T
        buf[o:o + T] = first.view(np.int64)
        buf[o_qn:o_qn + len(qn)] = qn  # sample sketch size = sum of slice sizes (one per SKETCHED k)
        buf[o_scal:o_scal + 2] = scalars.view(np.int64)

    def _read_reduce(self, buf, mm):
        G, T, W, K = self.G, self.T, self.world, len(self.ks)
        o_sizes, o_count, o_bases, o_first, o_qn, o_scal, _ = self._red_layout()
        hits = buf[:K * G].astype(np.uint32).reshape(K, G)
        sizes = buf[o_sizes:o_sizes + K * G].astype(np.uint32).reshape(K, G)
        count, bases = buf[o_count:o_count + T].view(np.uint64), buf[o_bases:o_bases + T].view(np.uint64)
        first = buf[o_first:o_first + W * T].view(np.uint64).reshape(W, T).min(axis=0)  # shards hold disjoint, increasing read-index ranges
        scalars = buf[o_scal:o_scal + 2].view(np.uint64)
        return self._pack_out(hits, sizes, count, bases, first, scalars, [int(x) for x in buf[o_qn:o_qn + len(self.sks_k)]], mm)

    def _pack_out(self, hits, sizes, count, bases, first, scalars, qn, mm):
        out = dict(hits_k=hits, sizes_k=sizes, hits=hits[-1], sizes=sizes[-1], count=count, bases=bases, first_seen=first,
                   tot_rds=int(scalars[0]), n_ambig=int(scalars[1]), sketch_sizes=qn, sketch_size=qn[-1], multimapped=mm,
                   ks=list(self.ks), sketched_ks=list(self.sks_k), definition=self.definition, match=getattr(self, "match", None))
        ci_vals = hits / np.maximum(sizes, 1)
        out["containment_k"] = ci_vals
        out["containment"] = ci_vals[-1]  # the largest k: the column the cutoff reads (select_db.py:85-86)
        out["top_ok"] = bool(ci_vals[-1].max() > 0.5) if ci_vals.shape[1] else None
        return out

    def _run_exchange_pipelined(self, nsteps, want_multimapped):
        eng, t, dist, W, K = self.engine, self.torch, self.dist, self.world, len(self.sks_k)
        G, T = self.G, self.T
        NSLOT, NW = 4, K * (W + 4) + 3  # per rank, per k: W slice sizes | truncated | last hash | n | overflows; then m0 | m1 | reads
        eng.x_setup(W, G, T, self.bounds, NSLOT)

        def gather_words(P, word_t):
            words = t.empty((W, NW), dtype=t.int64, device=word_t.device)
            dist.all_gather(list(words.unbind(0)), word_t)
            self._sent("words_all_gather", 8 * NW * (W - 1))
            eng.x_fetch_words(P, words, word_t)

        def phase_a(P, slot):  # this rank's words into the all-gather: no host wait at all
            gather_words(P, eng.x_words(P, slot))

        def phase_b(P):  # all-to-all of the slices, stage-C commit, merge, stage B
            words = eng.x_wait_words(P)
            tail = K * (W + 4)
            if any(w[ki * (W + 4) + W + 3] for w in words for ki in range(K)):
                # some rank's counting table overflowed (a sample unlike the previous one): its words are stale.  Every
                # rank sees the same flags, so every rank repeats the all-gather once that sketch has been rebuilt.
                self.words_redone = getattr(self, "words_redone", 0) + 1
                gather_words(P, eng.x_redo_words(P, self.bounds, words[self.rank][tail:]))
                words = eng.x_wait_words(P)
            received, inflight = [], []
            for ki, sk in enumerate(P["sks"] if self.match != "kmer" else []):
                o = ki * (W + 4)
                sc = [int(x) for x in words[self.rank][o:o + W]]
                recv_counts = [int(words[p][o + self.rank]) for p in range(W)]
                h, c = eng.export_sketch(sk)
                rh, rc, fl = self._all_to_all(h, c, sc, recv_counts)
                received.append((rh, rc))
                inflight += fl
            maps = [(w[tail], w[tail + 1]) for w in words]
            incoming = compose_incoming(maps, self.rank)
            group_base = int(sum(w[tail + 2] for w in words[: self.rank]))
            first_shard = self.nonempty[self.rank] and not any(self.nonempty[: self.rank])
            eng.x_commit(P, incoming, first_shard, group_base)  # runs during the all-to-all: it needs the maps only
            if self.match == "kmer":
                P["keep"] = self._sum_kmer_counts(P["sks"][0].counts)
                eng.x_stage_b(P, P["sks"], self.ci)
                return
            for wk in inflight:
                wk.wait()
            merged = []
            for ki, (rh, rc) in enumerate(received):
                any_trunc, complete_to, lo, hi = self._merge_args(words, ki)
                # queued, not waited for: the merged slice is consumed on the device by stage B; phase C settles it
                m = eng.x_merge(P, rh, rc, self.sks_k[ki], lo, hi, any_trunc, complete_to)
                if self.s or any_trunc:
                    m = self._bottom_s(m, any_trunc, self.sks_k[ki])
                merged.append(m)
            eng.x_stage_b(P, merged, self.ci)

        def phase_c(P):  # this rank's counts -> THE all-reduce
            hits, sizes, count, bases, first, scalars, P["mm"], qn = eng.x_collect(P, self.ci, want_multimapped)
            buf = eng.x_reduce_buffer(P)
            self._fill_reduce(buf, hits, sizes, count, bases, first, scalars, qn)
            tb = eng.x_reduce_tensor(P)
            dist.all_reduce(tb, op=dist.ReduceOp.SUM)
            self._sent("results_all_reduce", int(tb.numel()) * tb.element_size() if W > 1 else 0)
            self.traffic["passes"] += 1
            eng.x_fetch_reduced(P, tb)

        def phase_d(P):
            return self._read_reduce(eng.x_wait_reduced(P), P["mm"])

        import gc
        gc_was_on = gc.isenabled()
        gc.disable()  # a generation-2 collection in the middle of a tick is a 2-3 ms hole in the GPU's queue (measured:
        eng.x_begin()  # one per ~160 passes); nothing cyclic is created here, reference counting frees what a pass drops
        try:
            AHEAD = 3  # stage A queued this many passes ahead: the GPU keeps hashing through a host stall of a millisecond or two

            def front():
                return eng.x_front(self.sks_k, self.hmaxs, self.s, self.pct_id)
            fronts = [front() for _ in range(min(AHEAD, nsteps))]
            passes, out = {}, None
            for tick in range(nsteps + 3):
                if 0 <= tick - 3 < nsteps:
                    out = phase_d(passes.pop(tick - 3))
                if 0 <= tick - 2 < nsteps:
                    phase_c(passes[tick - 2])
                if 0 <= tick - 1 < nsteps:
                    phase_b(passes[tick - 1])
                # phase A last: it makes the main stream wait (on the device) for the NEWEST sketch, and whatever is
                # queued behind that wait only runs once that hashing kernel is through
                if tick < nsteps:
                    passes[tick] = P = fronts.pop(0)
                    phase_a(P, tick % NSLOT)
                    if tick + AHEAD < nsteps:
                        fronts.append(front())
            return out
        finally:
            eng.x_end()
            if gc_was_on:
                gc.enable()

    def step(self, want_multimapped=False, _sketch=None):
        """One pass of the hot path over the resident batch.  Returns the sample-wide results (every rank).
        _sketch: this pass's stage A, already queued (run())."""
        eng = self.engine
        self._want_mm = want_multimapped
        self._given = _sketch
        if self.exchange:
            sks, committed = self._exchange_step()
            if committed is None:  # commit is in flight: one read-back for stage B's counts and its accumulators
                (hits, sizes), committed = eng.containment_and_commit_results(sks, self.ci, want_multimapped)
            else:
                hits, sizes = eng.containment(sks, self.ci)
        else:
            # single shard: stage C is queued first, its results come back with the containment counts ...
            split = hasattr(eng, "profile_commit_launch")
            # stage A does not synchronise: the whole step is queued, then read back once.  Stage A is queued first
            # (its persistent grid takes the CUs); stage C follows on the second stream and fills in as stage A drains.
            if _sketch is not None:
                sks, ahead = _sketch
                ahead.free()  # (a single shard needs no map-only pass; its handle is created below)
            else:
                sks = (eng.sketch_local_async(self.sks_k, self.hmaxs, self.s) if split
                       else eng.sketch_local(self.sks_k, self.hmaxs, self.s))
            eng.profile_begin(self.pct_id, False)
            if split:
                eng.profile_commit_launch(1, True, 0)
            if split:
                (hits, sizes), committed = eng.containment_and_commit_results(sks, self.ci, want_multimapped)
            else:
                hits, sizes = eng.containment(sks, self.ci)
                committed = eng.profile_commit(1, True, 0, want_multimapped)
        return self._results(sks, hits, sizes, committed)

    def _results(self, sks, hits, sizes, committed):
        """Sample-wide results from this rank's stage B counts and stage C accumulators (the all-reduce when sharded)."""
        count, bases, first, scalars, mm = committed
        hits, sizes = np.asarray(hits), np.asarray(sizes)
        # (stage A by k-mer identity leaves no sketch: the matched pairs of the largest k stand in its place)
        qn = [sk.size if sk.size is not None else int(hits[-1].sum()) for sk in sks]
        for sk in sks:
            sk.free()
        if self.exchange:
            t, dist = self.torch, self.dist
            buf = np.zeros(self._red_layout()[-1], dtype=np.int64)
            self._fill_reduce(buf, hits, sizes, count, bases, first, scalars, qn)
            tb = t.as_tensor(buf, device=self.device)
            dist.all_reduce(tb, op=dist.ReduceOp.SUM)  # THE all-reduce
            self._sent("results_all_reduce", int(tb.numel()) * tb.element_size() if self.world > 1 else 0)
            self.traffic["passes"] += 1
            return self._read_reduce(tb.cpu().numpy(), mm)
        return self._pack_out(hits.astype(np.uint32), sizes.astype(np.uint32), count, bases, first, scalars, qn, mm)
