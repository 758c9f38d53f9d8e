"""CPU ORACLE — test infrastructure, not product code.

ctypes front end of oracle/libmg_oracle.so (built by oracle/Makefile) plus a
pure-Python restatement of the SAM line filter used at ingest.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; metalign_amd/ never does.

Pinning: see the header of oracle/mg_oracle.c and DESIGN.md.  Stage C and
MurmurHash3 are pinned by golden vectors; the sketch / containment arithmetic
is "parity unpinned" (KMC 3 and CMash are absent from the reference tree).

Reference paths below are relative to /root/reference.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

REC_DTYPE = np.dtype([("ref_new", "<u4"), ("matched", "<u4"), ("total", "<u4"), ("flag_len", "<u4")])
NEW_BIT = 0x80000000
U64_MAX = 0xFFFFFFFFFFFFFFFF


def build():
    """Compile the C restatement (gcc); idempotent."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "libmg_oracle.so"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmg_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def murmur3_x64_128(data: bytes, seed: int = 0):
    out = (ctypes.c_uint64 * 2)()
    lib().mgo_murmur3_x64_128(ctypes.c_char_p(data), ctypes.c_int(len(data)), ctypes.c_uint32(seed), out)
    return int(out[0]), int(out[1])


def kmer_hashes(seq: bytes, k: int):
    """(hashes, valid) per window start of one sequence."""
    n = max(len(seq) - k + 1, 0)
    out = np.zeros(n, dtype=np.uint64)
    valid = np.zeros(n, dtype=np.uint8)
    buf = np.frombuffer(seq, dtype=np.uint8)
    if n:
        lib().mgo_kmer_hashes.restype = ctypes.c_uint64
        lib().mgo_kmer_hashes(_p(buf, ctypes.c_uint8), ctypes.c_uint64(len(seq)), ctypes.c_int(k),
                              _p(out, ctypes.c_uint64), _p(valid, ctypes.c_uint8))
    return out, valid.astype(bool)


CMASH_PRIME = 9999999999971


def set_hash_mode(mode):
    """0: hash(min(kmer, revcomp)), 64 bits (default); 1: min(hash(kmer), hash(revcomp)) % CMASH_PRIME (CMash as
    SURVEY.md §8c recollects it, unverified).  Process-wide, like the library's mg_set_hash_mode."""
    lib().mgo_set_hash_mode(ctypes.c_int(int(mode)))


def hash_mode():
    return int(lib().mgo_hash_mode())


DEFAULT_CS = 3  # kmc -cs3 (scripts/select_db.py:50): occurrence counters saturate at 3; the library's default too


def sketch_reads(bases, offsets, k, hmax=U64_MAX, s=0, cap=None, cs=DEFAULT_CS):
    """-> (hashes u64[n], counts u32[n], truncated, kmers_seen); counts = min(occurrences, cs) (cs = 0: exact)."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    nreads = len(offsets) - 1
    if cap is None:
        cap = int(s) if s else max(int(bases.size), 1)
    h = np.zeros(cap, dtype=np.uint64)
    c = np.zeros(cap, dtype=np.uint32)
    n = ctypes.c_uint64(0)
    trunc = ctypes.c_int(0)
    seen = ctypes.c_uint64(0)
    bptr = _p(bases, ctypes.c_uint8) if bases.size else ctypes.POINTER(ctypes.c_uint8)()
    rc = lib().mgo_sketch_reads(bptr, _p(offsets, ctypes.c_uint64), ctypes.c_uint64(nreads), ctypes.c_int(k),
                                ctypes.c_uint64(hmax), ctypes.c_uint64(s), ctypes.c_uint32(cs), _p(h, ctypes.c_uint64),
                                _p(c, ctypes.c_uint32), ctypes.c_uint64(cap), ctypes.byref(n),
                                ctypes.byref(trunc), ctypes.byref(seen))
    if rc != 0:
        raise RuntimeError("mgo_sketch_reads rc=%d" % rc)
    return h[: n.value].copy(), c[: n.value].copy(), bool(trunc.value), int(seen.value)


def filter_bits(hashes):
    """Membership pre-filter of include/metalign_hip.h (mg_filter_build): -> (bool array of 2^b bits, mask)."""
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    lb = 16
    while lb < 30 and (1 << lb) < 16 * len(hashes):
        lb += 1
    bits = np.zeros(1 << lb, dtype=bool)
    mask = np.uint64((1 << lb) - 1)
    bits[(hashes & mask).astype(np.int64)] = True
    return bits, mask


def sketch_reads_filtered(bases, offsets, k, table_hashes, hmax=U64_MAX, s=0, cs=DEFAULT_CS):
    """The filtered read sketch: the sketch of sketch_reads restricted to hashes whose filter bit is set, THEN cut to
    the s smallest.  -> (hashes, counts, truncated, kmers_seen)."""
    h, c, _, seen = sketch_reads(bases, offsets, k, hmax=hmax, s=0, cs=cs)
    bits, mask = filter_bits(table_hashes)
    keep = bits[(h & mask).astype(np.int64)] if len(h) else np.zeros(0, dtype=bool)
    h, c = h[keep], c[keep]
    truncated = bool(s) and len(h) > s
    if truncated:
        h, c = h[:s], c[:s]
    return h, c, truncated, seen


def sketch_genomes(bases, offsets, k, n):
    """-> (hashes u64[*], offsets u64[G+1])."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    g = len(offsets) - 1
    h = np.zeros(max(g * n, 1), dtype=np.uint64)
    o = np.zeros(g + 1, dtype=np.uint64)
    rc = lib().mgo_sketch_genomes(_p(bases, ctypes.c_uint8), _p(offsets, ctypes.c_uint64), ctypes.c_uint64(g),
                                  ctypes.c_int(k), ctypes.c_uint64(n), _p(h, ctypes.c_uint64),
                                  _p(o, ctypes.c_uint64))
    if rc != 0:
        raise RuntimeError("mgo_sketch_genomes rc=%d" % rc)
    return h[: int(o[-1])].copy(), o


def sketch_genomes_prefix(bases, offsets, kmax, k, n):
    """The k < kmax table of hash mode 1: per genome the distinct mode-1 hashes of the k-prefixes of its sketched kmax-mers
    (oracle/mg_oracle.c: mgo_sketch_genomes_prefix).  -> (hashes u64[*], offsets u64[G+1])."""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    g = len(offsets) - 1
    h = np.zeros(max(g * n, 1), dtype=np.uint64)
    o = np.zeros(g + 1, dtype=np.uint64)
    rc = lib().mgo_sketch_genomes_prefix(_p(bases, ctypes.c_uint8), _p(offsets, ctypes.c_uint64), ctypes.c_uint64(g),
                                         ctypes.c_int(kmax), ctypes.c_int(k), ctypes.c_uint64(n), _p(h, ctypes.c_uint64),
                                         _p(o, ctypes.c_uint64))
    if rc != 0:
        raise RuntimeError("mgo_sketch_genomes_prefix rc=%d" % rc)
    return h[: int(o[-1])].copy(), o


def sketch_genomes_kmers(bases, offsets, k, n, sketch_hash="canonical"):
    """Stage A' with the k-mers kept (oracle/mg_oracle.c: mgo_sketch_genomes_kmers), under the mode in force.
    -> (hashes u64[*], kmer_hi u64[*], kmer_lo u64[*], offsets u64[G+1]); the k-mer 2-bit packed, first base most significant.
    sketch_hash = "forward": mgo_sketch_genomes_kmers_forward (entries selected by the forward k-mer's hash, kept as they stand)."""
    fn = lib().mgo_sketch_genomes_kmers_forward if sketch_hash == "forward" else lib().mgo_sketch_genomes_kmers
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    g = len(offsets) - 1
    h = np.zeros(max(g * n, 1), dtype=np.uint64)
    hi = np.zeros(max(g * n, 1), dtype=np.uint64)
    lo = np.zeros(max(g * n, 1), dtype=np.uint64)
    o = np.zeros(g + 1, dtype=np.uint64)
    rc = fn(_p(bases, ctypes.c_uint8), _p(offsets, ctypes.c_uint64), ctypes.c_uint64(g), ctypes.c_int(k), ctypes.c_uint64(n),
            _p(h, ctypes.c_uint64), _p(hi, ctypes.c_uint64), _p(lo, ctypes.c_uint64), _p(o, ctypes.c_uint64))
    if rc != 0:
        raise RuntimeError("mgo_sketch_genomes_kmers rc=%d" % rc)
    e = int(o[-1])
    return h[:e].copy(), hi[:e].copy(), lo[:e].copy(), o


def refpipe_build(hashes, kmer_hi, kmer_lo, offsets, ks):
    """The table of the reference pipeline (oracle/mg_oracle.c, "THE REFERENCE'S OWN WIRING"): from the genome-major
    entries of the largest k (sketch_genomes_kmers) -> dict(ks, ngenomes, pair_hash, pair_gen, gsize, kmer_hi, kmer_lo (pair
    order), small = {k: dict(pa, pb, cid, cgen, gsize, nprefix)} for every k below the largest)."""
    hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    ks = [int(k) for k in ks]
    kmax, g, e = ks[-1], len(offsets) - 1, len(hashes)
    ph = np.zeros(max(e, 1), dtype=np.uint64)
    pg = np.zeros(max(e, 1), dtype=np.uint32)
    perm = np.zeros(max(e, 1), dtype=np.uint64)
    rc = lib().mgo_refpipe_pairs(_p(hashes, ctypes.c_uint64), _p(offsets, ctypes.c_uint64), ctypes.c_uint64(g),
                                 _p(ph, ctypes.c_uint64), _p(pg, ctypes.c_uint32), _p(perm, ctypes.c_uint64))
    if rc != 0:
        raise RuntimeError("mgo_refpipe_pairs rc=%d" % rc)
    ph, pg, perm = ph[:e], pg[:e], perm[:e].astype(np.int64)
    khi = np.ascontiguousarray(np.asarray(kmer_hi, dtype=np.uint64)[perm])
    klo = np.ascontiguousarray(np.asarray(kmer_lo, dtype=np.uint64)[perm])
    out = dict(ks=ks, ngenomes=g, pair_hash=ph, pair_gen=pg, gsize=np.diff(offsets).astype(np.uint32), kmer_hi=khi,
               kmer_lo=klo, small={})
    for k in ks[:-1]:
        pa, pb = np.zeros(max(e, 1), np.uint32), np.zeros(max(e, 1), np.uint32)
        cid, cgen = np.zeros(max(e, 1), np.uint32), np.zeros(max(e, 1), np.uint32)
        gs = np.zeros(max(g, 1), np.uint32)
        nc, npre = ctypes.c_uint64(0), ctypes.c_uint64(0)
        rc = lib().mgo_refpipe_build_k(_p(khi if e else np.zeros(1, np.uint64), ctypes.c_uint64),
                                       _p(klo if e else np.zeros(1, np.uint64), ctypes.c_uint64),
                                       _p(np.ascontiguousarray(pg) if e else np.zeros(1, np.uint32), ctypes.c_uint32),
                                       ctypes.c_uint64(e), ctypes.c_uint64(g), ctypes.c_int(kmax), ctypes.c_int(k),
                                       _p(pa, ctypes.c_uint32), _p(pb, ctypes.c_uint32), _p(cid, ctypes.c_uint32),
                                       _p(cgen, ctypes.c_uint32), ctypes.byref(nc), _p(gs, ctypes.c_uint32), ctypes.byref(npre))
        if rc != 0:
            raise RuntimeError("mgo_refpipe_build_k rc=%d" % rc)
        out["small"][k] = dict(pa=pa[:e].copy(), pb=pb[:e].copy(), cid=cid[:nc.value].copy(), cgen=cgen[:nc.value].copy(),
                               gsize=gs[:g].copy(), nprefix=int(npre.value))
    return out


def refpipe_containment(q_hashes, q_counts, ci, table):
    """The query of the reference pipeline against refpipe_build's table: q = the read sketch of the LARGEST k.
    -> (hits u32[K][G], sizes u32[K][G]), k ascending (the largest k last: the column the cutoff reads)."""
    q_hashes = np.ascontiguousarray(q_hashes, dtype=np.uint64)
    q_counts = np.ascontiguousarray(q_counts, dtype=np.uint32)
    ph, pg, g = table["pair_hash"], table["pair_gen"], table["ngenomes"]
    e = len(ph)
    matched = np.zeros(max(e, 1), dtype=np.uint8)
    lib().mgo_refpipe_matched(_p(q_hashes if len(q_hashes) else np.zeros(1, np.uint64), ctypes.c_uint64),
                              _p(q_counts if len(q_counts) else np.zeros(1, np.uint32), ctypes.c_uint32),
                              ctypes.c_uint64(len(q_hashes)), ctypes.c_uint32(ci),
                              _p(np.ascontiguousarray(ph) if e else np.zeros(1, np.uint64), ctypes.c_uint64), ctypes.c_uint64(e),
                              _p(matched, ctypes.c_uint8))
    hits, sizes = [], []
    for k in table["ks"][:-1]:
        t = table["small"][k]
        out = np.zeros(max(g, 1), dtype=np.uint32)
        rc = lib().mgo_refpipe_hits_k(_p(matched, ctypes.c_uint8),
                                      _p(t["pa"] if e else np.zeros(1, np.uint32), ctypes.c_uint32),
                                      _p(t["pb"] if e else np.zeros(1, np.uint32), ctypes.c_uint32), ctypes.c_uint64(e),
                                      ctypes.c_uint64(t["nprefix"]),
                                      _p(t["cid"] if len(t["cid"]) else np.zeros(1, np.uint32), ctypes.c_uint32),
                                      _p(t["cgen"] if len(t["cgen"]) else np.zeros(1, np.uint32), ctypes.c_uint32),
                                      ctypes.c_uint64(len(t["cid"])), ctypes.c_uint64(g), _p(out, ctypes.c_uint32))
        if rc != 0:
            raise RuntimeError("mgo_refpipe_hits_k rc=%d" % rc)
        hits.append(out[:g].copy())
        sizes.append(t["gsize"].copy())
    hits.append(np.bincount(pg[matched[:e] != 0], minlength=g).astype(np.uint32)[:g] if e else np.zeros(g, np.uint32))
    sizes.append(table["gsize"].copy())
    return np.asarray(hits, dtype=np.uint32).reshape(len(table["ks"]), g), np.asarray(sizes, dtype=np.uint32).reshape(len(table["ks"]), g)


def refpipe_count_kmers(bases, offsets, k, kmer_hi, kmer_lo, cs=DEFAULT_CS):
    """Stage A of the reference pipeline by k-mer IDENTITY (oracle/mg_oracle.c: mgo_refpipe_count_kmers): per pair of a table
    (kmer_hi / kmer_lo in pair order) the occurrences of its canonical k-mer among the reads' canonical k-mers — what `kmc` +
    `kmc_tools intersect` compute (scripts/select_db.py:50-59) — saturating at cs (0: exact).  -> (counts u32[npairs], kmers_seen)"""
    bases = np.ascontiguousarray(bases, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    khi = np.ascontiguousarray(kmer_hi, dtype=np.uint64)
    klo = np.ascontiguousarray(kmer_lo, dtype=np.uint64)
    e = len(khi)
    out = np.zeros(max(e, 1), dtype=np.uint32)
    seen = ctypes.c_uint64(0)
    bptr = _p(bases, ctypes.c_uint8) if bases.size else ctypes.POINTER(ctypes.c_uint8)()
    rc = lib().mgo_refpipe_count_kmers(bptr, _p(offsets, ctypes.c_uint64), ctypes.c_uint64(len(offsets) - 1), ctypes.c_int(int(k)),
                                       ctypes.c_uint32(int(cs)), _p(khi if e else np.zeros(1, np.uint64), ctypes.c_uint64),
                                       _p(klo if e else np.zeros(1, np.uint64), ctypes.c_uint64), ctypes.c_uint64(e),
                                       _p(out, ctypes.c_uint32), ctypes.byref(seen))
    if rc != 0:
        raise RuntimeError("mgo_refpipe_count_kmers rc=%d" % rc)
    return out[:e].copy(), int(seen.value)


class KmerTable:
    """The table's canonical k-mers as a look-up set, built once (mgo_kmer_table_new); count() may run on several threads at once
    (ctypes releases the GIL), each call with counters of its own — bench.py's cpu_baseline."""

    def __init__(self, kmer_hi, kmer_lo, k):
        khi = np.ascontiguousarray(kmer_hi, dtype=np.uint64)
        klo = np.ascontiguousarray(kmer_lo, dtype=np.uint64)
        self.npairs, self.k = len(khi), int(k)
        one = np.zeros(1, np.uint64)
        lib().mgo_kmer_table_new.restype = ctypes.c_void_p
        self.handle = ctypes.c_void_p(lib().mgo_kmer_table_new(_p(khi if self.npairs else one, ctypes.c_uint64),
                                                               _p(klo if self.npairs else one, ctypes.c_uint64),
                                                               ctypes.c_uint64(self.npairs), ctypes.c_int(self.k)))
        if not self.handle:
            raise MemoryError("mgo_kmer_table_new")

    def count(self, bases, offsets):
        """-> (uint64[npairs] exact occurrences, at the first pair of every k-mer; kmers_seen)"""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        cnt = np.zeros(max(self.npairs, 1), dtype=np.uint64)
        seen = ctypes.c_uint64(0)
        bptr = _p(bases, ctypes.c_uint8) if bases.size else ctypes.POINTER(ctypes.c_uint8)()
        rc = lib().mgo_kmer_table_count(self.handle, bptr, _p(offsets, ctypes.c_uint64), ctypes.c_uint64(len(offsets) - 1),
                                        _p(cnt, ctypes.c_uint64), ctypes.byref(seen))
        if rc != 0:
            raise RuntimeError("mgo_kmer_table_count rc=%d" % rc)
        return cnt, int(seen.value)

    def per_pair(self, counts, cs=DEFAULT_CS):
        """counts: the sum of count()'s arrays over the sample's shares -> uint32[npairs]: min(occurrences of the pair's k-mer, cs)"""
        counts = np.ascontiguousarray(counts, dtype=np.uint64)
        out = np.zeros(max(self.npairs, 1), dtype=np.uint32)
        rc = lib().mgo_kmer_table_per_pair(self.handle, _p(counts, ctypes.c_uint64), ctypes.c_uint32(int(cs)), _p(out, ctypes.c_uint32))
        if rc != 0:
            raise RuntimeError("mgo_kmer_table_per_pair rc=%d" % rc)
        return out[: self.npairs].copy()

    def free(self):
        if self.handle:
            lib().mgo_kmer_table_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


def refpipe_containment_counts(counts, ci, table):
    """refpipe_containment with the matched pairs taken from per-pair occurrence counts (refpipe_count_kmers): counts >= ci."""
    ph, pg, g = table["pair_hash"], table["pair_gen"], table["ngenomes"]
    e = len(ph)
    matched = np.zeros(max(e, 1), dtype=np.uint8)
    matched[:e] = np.asarray(counts, dtype=np.uint32)[:e] >= ci
    hits, sizes = [], []
    for k in table["ks"][:-1]:
        t = table["small"][k]
        out = np.zeros(max(g, 1), dtype=np.uint32)
        rc = lib().mgo_refpipe_hits_k(_p(matched, ctypes.c_uint8),
                                      _p(t["pa"] if e else np.zeros(1, np.uint32), ctypes.c_uint32),
                                      _p(t["pb"] if e else np.zeros(1, np.uint32), ctypes.c_uint32), ctypes.c_uint64(e),
                                      ctypes.c_uint64(t["nprefix"]),
                                      _p(t["cid"] if len(t["cid"]) else np.zeros(1, np.uint32), ctypes.c_uint32),
                                      _p(t["cgen"] if len(t["cgen"]) else np.zeros(1, np.uint32), ctypes.c_uint32),
                                      ctypes.c_uint64(len(t["cid"])), ctypes.c_uint64(g), _p(out, ctypes.c_uint32))
        if rc != 0:
            raise RuntimeError("mgo_refpipe_hits_k rc=%d" % rc)
        hits.append(out[:g].copy())
        sizes.append(t["gsize"].copy())
    hits.append(np.bincount(pg[matched[:e] != 0], minlength=g).astype(np.uint32)[:g] if e else np.zeros(g, np.uint32))
    sizes.append(table["gsize"].copy())
    return np.asarray(hits, dtype=np.uint32).reshape(len(table["ks"]), g), np.asarray(sizes, dtype=np.uint32).reshape(len(table["ks"]), g)


def refpipe_matched(q_hashes, q_counts, ci, pair_hash):
    """uint8[npairs]: the pair's hash is in the read sketch with count >= ci (mgo_refpipe_matched)."""
    q_hashes = np.ascontiguousarray(q_hashes, dtype=np.uint64)
    q_counts = np.ascontiguousarray(q_counts, dtype=np.uint32)
    ph = np.ascontiguousarray(pair_hash, dtype=np.uint64)
    matched = np.zeros(max(len(ph), 1), dtype=np.uint8)
    lib().mgo_refpipe_matched(_p(q_hashes if len(q_hashes) else np.zeros(1, np.uint64), ctypes.c_uint64),
                              _p(q_counts if len(q_counts) else np.zeros(1, np.uint32), ctypes.c_uint32),
                              ctypes.c_uint64(len(q_hashes)), ctypes.c_uint32(ci),
                              _p(ph if len(ph) else np.zeros(1, np.uint64), ctypes.c_uint64), ctypes.c_uint64(len(ph)),
                              _p(matched, ctypes.c_uint8))
    return matched[: len(ph)]


def refpipe_mark_words(matched, t):
    """The prefix bitmap a set of matched pairs marks (numpy; bit p of word p >> 5 = prefix p): t = dict(pa, pb, nprefix) of one
    k below the largest, for the pairs `matched` is about.  -> uint32[(nprefix + 31) // 32]"""
    nw = (int(t["nprefix"]) + 31) // 32
    bits = np.zeros(nw * 32, dtype=np.uint8)
    m = np.asarray(matched, dtype=bool)
    bits[np.asarray(t["pa"])[m].astype(np.int64)] = 1
    pb = np.asarray(t["pb"])[m]
    bits[pb[pb != 0xFFFFFFFF].astype(np.int64)] = 1
    return np.packbits(bits, bitorder="little").view("<u4").copy() if nw else np.zeros(0, np.uint32)


def refpipe_count_words(words, t, ngenomes):
    """hits u32[G] of one k below the largest from a prefix bitmap and (a run of) the k's count list t = dict(cid, cgen)."""
    words = np.ascontiguousarray(words, dtype="<u4")
    bits = np.unpackbits(words.view(np.uint8), bitorder="little") if len(words) else np.zeros(0, np.uint8)
    cid, cgen = np.asarray(t["cid"]).astype(np.int64), np.asarray(t["cgen"]).astype(np.int64)
    on = bits[cid] == 1 if len(cid) else np.zeros(0, bool)
    return np.bincount(cgen[on], minlength=ngenomes).astype(np.uint32)[:ngenomes]


def containment(q_hashes, q_counts, q_truncated, ci, db_hashes, db_offsets):
    """-> (hits u32[G], sizes u32[G])."""
    q_hashes = np.ascontiguousarray(q_hashes, dtype=np.uint64)
    q_counts = np.ascontiguousarray(q_counts, dtype=np.uint32)
    db_hashes = np.ascontiguousarray(db_hashes, dtype=np.uint64)
    db_offsets = np.ascontiguousarray(db_offsets, dtype=np.uint64)
    g = len(db_offsets) - 1
    hits = np.zeros(g, dtype=np.uint32)
    sizes = np.zeros(g, dtype=np.uint32)
    lib().mgo_containment(_p(q_hashes, ctypes.c_uint64), _p(q_counts, ctypes.c_uint32),
                          ctypes.c_uint64(len(q_hashes)), ctypes.c_int(int(q_truncated)), ctypes.c_uint32(ci),
                          _p(db_hashes, ctypes.c_uint64), _p(db_offsets, ctypes.c_uint64), ctypes.c_uint64(g),
                          _p(hits, ctypes.c_uint32), _p(sizes, ctypes.c_uint32))
    return hits, sizes


def profile_assign(recs, ref2tax, ntax, pct_id):
    """Sequential restatement of map_and_process over records.

    -> dict(count, bases, first_seen, tot_rds, n_ambig, mm_offsets, mm_tax, mm_hitlen, mm_read)
    """
    recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
    ref2tax = np.ascontiguousarray(ref2tax, dtype=np.uint32)
    n = len(recs)
    count = np.zeros(ntax, dtype=np.uint64)
    bases = np.zeros(ntax, dtype=np.uint64)
    first = np.zeros(ntax, dtype=np.uint64)
    mm_off = np.zeros(n + 2, dtype=np.uint64)
    mm_tax = np.zeros(n + 1, dtype=np.uint32)
    mm_len = np.zeros(n + 1, dtype=np.uint64)
    mm_read = np.zeros(n + 1, dtype=np.uint64)
    tot = ctypes.c_uint64(0)
    amb = ctypes.c_uint64(0)
    nmm = ctypes.c_uint64(0)
    nent = ctypes.c_uint64(0)
    rc = lib().mgo_profile_assign(
        recs.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(n), _p(ref2tax, ctypes.c_uint32),
        ctypes.c_uint32(len(ref2tax)), ctypes.c_uint32(ntax), ctypes.c_double(pct_id),
        _p(count, ctypes.c_uint64), _p(bases, ctypes.c_uint64), _p(first, ctypes.c_uint64),
        ctypes.byref(tot), ctypes.byref(amb), _p(mm_off, ctypes.c_uint64), _p(mm_tax, ctypes.c_uint32),
        _p(mm_len, ctypes.c_uint64), _p(mm_read, ctypes.c_uint64), ctypes.c_uint64(n + 1),
        ctypes.c_uint64(n + 1), ctypes.byref(nmm), ctypes.byref(nent))
    if rc != 0:
        raise RuntimeError("mgo_profile_assign rc=%d" % rc)
    m, e = nmm.value, nent.value
    return dict(count=count, bases=bases, first_seen=first, tot_rds=tot.value, n_ambig=amb.value,
                mm_offsets=mm_off[: m + 1].copy(), mm_tax=mm_tax[:e].copy(), mm_hitlen=mm_len[:m].copy(),
                mm_read=mm_read[:m].copy())


# --------------------------------------------------------------------------
# SAM text -> records, the plain way (line filter of
# scripts/map_and_profile.py:201-217 and the CIGAR walk of :86-100).
# --------------------------------------------------------------------------
def sam_to_records(lines, acc_index):
    """lines: iterable of str; acc_index: {accession: row}. -> REC_DTYPE array.

    Raises what the reference raises on the same input: KeyError for an unknown
    RNAME (:217), IndexError for a retained line with < 12 fields (:97),
    ValueError for a CIGAR with '=' (:90-93).
    """
    out = []
    prev = ""
    for line in lines:
        if line.startswith("@"):
            continue
        f = line.strip().split()
        if len(f) < 6:
            continue
        flag = int(f[1])
        if (flag & 4) or f[5] == "*":
            continue
        ref = acc_index[f[2]]
        matched = total = cur = 0
        for ch in f[5]:
            if not ch.isalpha():
                cur = cur * 10 + int(ch)
            else:
                if ch == "M" or ch == "=":
                    matched += cur
                total += cur
                cur = 0
        int(f[11][5:])  # parsed and unused by the reference; still raises on short / malformed lines
        float(matched) / float(total)  # ZeroDivisionError parity for op-less CIGARs
        seqlen = 0 if f[9] == "*" else len(f[9])
        new = f[0] != prev
        prev = f[0]
        out.append((ref | (NEW_BIT if new else 0), matched, total, (flag & 0xFFF) | (seqlen << 12)))
    return np.array(out, dtype=REC_DTYPE) if out else np.zeros(0, dtype=REC_DTYPE)
