"""Data formats either side of the hot path: reads (FASTA / FASTQ, optionally gzipped) in, and the
on-disk genome sketch table that replaces the reference's CMash artefacts.

Sketch table directory (replaces data/cmash_db_n1000_k60.h5, ..._dump.kmc_{pre,suf} and
..._30-60-10.bf, /root/reference/scripts/select_db.py:44,69-70), version 2 — HASH-MAJOR, the layout stage B streams
(mg_contain.hip), so that nothing is sorted at load time and a rank of a multi-GPU job maps only its hash range:

    <dir>/meta.json             {"format": "metalign_amd.sketch_table", "version": 2, "n": 1000,
                                 "ks": [21, 31, 51], "ngenomes": G, "hash": "murmur3_x64_128.h1(canonical ASCII k-mer)"}
    <dir>/names.txt             one organism file name per line (taxid_<a>_<b>_genomic.fna.gz,
                                 /root/reference/utils/ncbi2db.py:160-186), row order == genome id
    <dir>/k<K>.pair_hash.u64    little-endian u64: every (hash, genome) pair of every genome sketch, ascending by hash
                                 (equal hashes of different genomes adjacent, in genome order)
    <dir>/k<K>.pair_gen.u32     little-endian u32: the genome of every pair
    <dir>/k<K>.gsize.u32        little-endian u32[G]: sketch size of every genome
    <dir>/k<K>.filter.u32       the membership pre-filter over all hashes of the table (one bit per hash value modulo
                                 the size, include/metalign_hip.h: mg_filter) — the reference's ..._30-60-10.bf
Version 1 (genome-major k<K>.hashes.u64 + k<K>.offsets.u64: every genome's ascending sketch back to back) is still
read; it is inverted on the host when it is opened.
Version 3 — the REFERENCE PIPELINE's table (meta "stage_a_definition": "reference_pipeline"; include/metalign_hip.h, mg_refdb):
the reads are sketched at the LARGEST k only, as the reference's kmc call does (select_db.py:50-52), and every smaller k's column
comes from the k-prefixes of the matched k_max-mers (the streaming query, :73-76).  The largest k has the version-2 files plus
    <dir>/k<KMAX>.kmer_hi.u64 / .kmer_lo.u64   the sketched k_max-mer of every pair as the table keeps it, 2-bit packed, first base
                                 most significant (CMash's database holds the sketches' k-mers too: local_tests/dump_kmers.py:7-14)
and every k below it has, instead of a table of its own,
    <dir>/k<K>.rp_pa.u32 / .rp_pb.u32          per pair of the k_max table: the number of the k-prefix of its k-mer / of the reverse
                                 complement's (0xffffffff: not a prefix of the table), numbered in ascending order of the prefixes
    <dir>/k<K>.rp_cid.u32 / .rp_cgen.u32       the distinct (prefix number, genome) combinations, ascending
    <dir>/k<K>.gsize.u32                       distinct k-prefixes per genome (the column's denominators)
    meta "nprefix": {K: number of distinct k-prefixes in the table}.
Flat files so that a RefSeq-scale table (2.4 GB per k) is np.memmap'ed and uploaded slice by slice.
"""
import gzip
import json
import os

import numpy as np

TABLE_FORMAT = "metalign_amd.sketch_table"


def _open_text(path):
    if path.endswith(".gz"):
        return gzip.open(path, "rt")
    return open(path, "r")


def read_sequences(path, kind):
    """FASTA or FASTQ -> (bases u8[total], offsets u64[n+1], names[list]).  Sequences are kept as written
    (case, N); the kernels upper-case and split k-mers on non-ACGT (oracle/mg_oracle.c: base_code)."""
    names, chunks, lens = [], [], []
    with _open_text(path) as fh:
        if kind == "fastq":
            while True:
                head = fh.readline()
                if not head:
                    break
                if not head.strip():  # blank line between / after records
                    continue
                seq = fh.readline().rstrip("\r\n")
                fh.readline()
                fh.readline()
                names.append(head[1:].split()[0] if len(head) > 1 else "")
                chunks.append(seq)
                lens.append(len(seq))
        else:
            cur = None
            for line in fh:
                if line.startswith(">"):
                    if cur is not None:
                        s = "".join(cur)
                        chunks.append(s)
                        lens.append(len(s))
                    names.append(line[1:].split()[0] if len(line) > 1 else "")
                    cur = []
                elif cur is not None:
                    cur.append(line.strip())
            if cur is not None:
                s = "".join(cur)
                chunks.append(s)
                lens.append(len(s))
    offsets = np.zeros(len(lens) + 1, dtype=np.uint64)
    if lens:
        offsets[1:] = np.cumsum(np.asarray(lens, dtype=np.uint64))
    bases = np.frombuffer("".join(chunks).encode("ascii", "replace"), dtype=np.uint8)
    return bases, offsets, names


class SketchTable:
    """Host view of a sketch table directory (version 2: hash-major; version 1: genome-major)."""

    def __init__(self, path):
        self.path = path
        with open(os.path.join(path, "meta.json")) as fh:
            self.meta = json.load(fh)
        if self.meta.get("format") != TABLE_FORMAT:
            raise ValueError("%s is not a %s directory" % (path, TABLE_FORMAT))
        self.version = int(self.meta.get("version", 1))
        self.ks = [int(k) for k in self.meta["ks"]]
        self.n = int(self.meta["n"])
        # which definition of a k-mer's hash the table was sketched with (include/metalign_hip.h: mg_set_hash_mode);
        # tables written before the field existed are mode 0
        self.hash_mode = int(self.meta.get("hash_mode", 0))
        # (version 3) what selected a genome's k-mers: "canonical" = the hash they match by; "forward" = build_db --sketch_hash forward
        self.sketch_hash = str(self.meta.get("sketch_hash", "canonical"))
        self.prefix_tables = bool(self.meta.get("prefix_tables", False))  # mode 1: k < k_max tables of k-prefixes (build_db)
        # "reference_pipeline": only the largest k is sketched on the read side; "sketch_per_k": every k has a table of its own
        self.definition = self.meta.get("stage_a_definition", "sketch_per_k")
        self.refpipe = self.definition == "reference_pipeline"
        with open(os.path.join(path, "names.txt")) as fh:
            self.names = fh.read().split("\n")
        if self.names and self.names[-1] == "":
            self.names.pop()
        self.ngenomes = len(self.names)
        self._maps = {}

    def _f(self, k, what):
        return os.path.join(self.path, "k%d.%s" % (k, what))

    def _pair_maps(self, k):
        if k not in self._maps:
            if self.version >= 2:
                ph = np.memmap(self._f(k, "pair_hash.u64"), dtype="<u8", mode="r")
                pg = np.memmap(self._f(k, "pair_gen.u32"), dtype="<u4", mode="r")
                gs = np.fromfile(self._f(k, "gsize.u32"), dtype="<u4")
            else:  # version 1: invert on the host (once per process)
                h, o = self.arrays(k)
                ph, pg, gs = pairs_from_genome_major(np.asarray(h), o)
            if len(ph) != len(pg) or len(gs) != self.ngenomes:
                raise ValueError("sketch table %s, k = %d: %d hashes, %d genome ids, %d genome sizes for %d genomes — the files do "
                                 "not belong together" % (self.path, k, len(ph), len(pg), len(gs), self.ngenomes))
            self._maps[k] = (ph, pg, gs)
        return self._maps[k]

    def max_hash(self, k):
        ph = self._pair_maps(k)[0]
        return int(ph[-1]) if len(ph) else 0

    def pairs(self, k, lo=None, hi=None):
        """The hash-major table of k restricted to hashes in [lo, hi) (default: all of it), as the keyword arguments of
        Hip.upload_table_sorted: pair_hash / pair_gen are views of the memory maps (only the slice is read from disk),
        gsize counts the pairs of every genome WITHIN the slice (a genome's containment is the sum over the slices)."""
        ph, pg, gs = self._pair_maps(k)
        mx = self.max_hash(k)
        if lo is None and hi is None:
            return dict(pair_hash=ph, pair_gen=pg, gsize=gs, max_hash=mx)
        a = int(np.searchsorted(ph, np.uint64(lo or 0), side="left"))
        b = len(ph) if hi is None or hi > 0xFFFFFFFFFFFFFFFF else int(np.searchsorted(ph, np.uint64(hi), side="left"))
        sl = np.bincount(np.asarray(pg[a:b]), minlength=self.ngenomes).astype(np.uint32)
        if len(sl) != self.ngenomes:  # (a genome id >= ngenomes in the slice: bincount grew)
            raise ValueError("sketch table %s, k = %d: genome id %d in a table of %d genomes" % (self.path, k, len(sl) - 1, self.ngenomes))
        return dict(pair_hash=ph[a:b], pair_gen=pg[a:b], gsize=sl, max_hash=mx)

    def refpipe_arrays(self):
        """The reference pipeline's table (version 3) as Hip.refdb_upload takes it — dict(ks, ngenomes, pair_hash, pair_gen, gsize,
        max_hash, small=[dict(pa, pb, cid, cgen, gsize, nprefix) per k below the largest]) — as memory maps.  A rank's share of it is
        cut by distributed.ShardJob (pairs by hash range, count lists by prefix range on multiples of 32: mark_cuts)."""
        if not self.refpipe:
            raise ValueError("%s is not a reference-pipeline table" % self.path)
        kmax = self.ks[-1]
        ph, pg, gs = self._pair_maps(kmax)
        small = []
        for k in self.ks[:-1]:
            npre = int(self.meta["nprefix"][str(k)])
            pa = np.memmap(self._f(k, "rp_pa.u32"), dtype="<u4", mode="r") if len(ph) else np.zeros(0, np.uint32)
            pb = np.memmap(self._f(k, "rp_pb.u32"), dtype="<u4", mode="r") if len(ph) else np.zeros(0, np.uint32)
            if os.path.getsize(self._f(k, "rp_cid.u32")):
                cid = np.memmap(self._f(k, "rp_cid.u32"), dtype="<u4", mode="r")
                cgen = np.memmap(self._f(k, "rp_cgen.u32"), dtype="<u4", mode="r")
            else:
                cid = cgen = np.zeros(0, np.uint32)
            gk = np.fromfile(self._f(k, "gsize.u32"), dtype="<u4")
            if len(pa) != len(ph) or len(pb) != len(ph) or len(cid) != len(cgen) or len(gk) != self.ngenomes:
                raise ValueError("sketch table %s, k = %d: the reference-pipeline files do not belong together" % (self.path, k))
            small.append(dict(pa=pa, pb=pb, cid=cid, cgen=cgen, gsize=gk, nprefix=npre))
        out = dict(ks=list(self.ks), ngenomes=self.ngenomes, pair_hash=ph, pair_gen=pg, gsize=gs,
                   max_hash=self.max_hash(kmax), small=small)
        # the sketched k_max-mers themselves, in pair order (what stage A by k-mer identity indexes; absent from tables written
        # without them)
        if os.path.exists(self._f(kmax, "kmer_hi.u64")) and os.path.exists(self._f(kmax, "kmer_lo.u64")):
            khi = np.memmap(self._f(kmax, "kmer_hi.u64"), dtype="<u8", mode="r") if len(ph) else np.zeros(0, np.uint64)
            klo = np.memmap(self._f(kmax, "kmer_lo.u64"), dtype="<u8", mode="r") if len(ph) else np.zeros(0, np.uint64)
            if len(khi) != len(ph) or len(klo) != len(ph):
                raise ValueError("sketch table %s, k = %d: the k-mer files do not belong to the pair list" % (self.path, kmax))
            out.update(kmer_hi=khi, kmer_lo=klo)
        return out

    def filter_bits(self, k):
        """The stored membership pre-filter of k (None for a table without one)."""
        f = self._f(k, "filter.u32")
        return np.fromfile(f, dtype="<u4") if os.path.exists(f) else None

    def arrays(self, k):
        """Genome-major view (hashes, offsets[G+1]): every genome's ascending sketch back to back."""
        if self.version >= 2:
            ph, pg, gs = self._pair_maps(k)
            order = np.argsort(np.asarray(pg), kind="stable")  # stable: ascending hash within a genome
            o = np.zeros(self.ngenomes + 1, dtype=np.uint64)
            o[1:] = np.cumsum(gs, dtype=np.uint64)
            return np.asarray(ph)[order], o
        h = np.memmap(self._f(k, "hashes.u64"), dtype="<u8", mode="r")
        o = np.fromfile(self._f(k, "offsets.u64"), dtype="<u8")
        assert len(o) == self.ngenomes + 1 and int(o[-1]) == len(h)
        return h, o


def pairs_from_genome_major(hashes, offsets):
    """(hashes, offsets[G+1]) genome-major -> (pair_hash ascending, pair_gen, gsize): the inversion mg_db_upload does on
    the device, on the host for the table builder."""
    hashes = np.asarray(hashes, dtype=np.uint64)
    offsets = np.asarray(offsets, dtype=np.uint64)
    g = len(offsets) - 1
    gsize = np.diff(offsets).astype(np.uint32)
    gen = np.repeat(np.arange(g, dtype=np.uint32), gsize)
    order = np.argsort(hashes, kind="stable")  # stable: equal hashes stay in genome order
    return hashes[order], gen[order], gsize


def _write_common(path, names):
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "names.txt"), "w") as fh:
        for nm in names:
            fh.write(nm + "\n")


def write_sketch_table(path, names, ks, n, per_k, filters=None, hash_mode=0, prefix_tables=False):
    """per_k: {k: (hashes u64[], offsets u64[G+1])} genome-major, as mg_sketch_genomes returns it; written hash-major.
    filters: optional {k: uint32 bit array} (Filter.download)."""
    _write_common(path, names)
    for k in ks:
        ph, pg, gs = pairs_from_genome_major(*per_k[k])
        np.ascontiguousarray(ph, dtype="<u8").tofile(os.path.join(path, "k%d.pair_hash.u64" % k))
        np.ascontiguousarray(pg, dtype="<u4").tofile(os.path.join(path, "k%d.pair_gen.u32" % k))
        np.ascontiguousarray(gs, dtype="<u4").tofile(os.path.join(path, "k%d.gsize.u32" % k))
        if filters and filters.get(k) is not None:
            np.ascontiguousarray(filters[k], dtype="<u4").tofile(os.path.join(path, "k%d.filter.u32" % k))
    meta = {"format": TABLE_FORMAT, "version": 2, "n": int(n), "ks": [int(k) for k in ks], "ngenomes": len(names), "hash_mode": int(hash_mode), "prefix_tables": bool(prefix_tables),
            "layout": "hash-major pairs", "hash": "murmur3_x64_128.h1(canonical ASCII k-mer), seed 0"}
    with open(os.path.join(path, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


def write_refpipe_table(path, names, n, table, filter_bits=None, hash_mode=0, sketch_hash="canonical"):
    """Version 3: the reference pipeline's table.  table: what _hip.RefTable.download() returns (dict(ks, pair_hash, pair_gen,
    gsize, kmer_hi, kmer_lo, small={k: dict(pa, pb, cid, cgen, gsize, nprefix)})); filter_bits: the membership pre-filter
    over the largest k's hashes (Filter.download)."""
    _write_common(path, names)
    ks = [int(k) for k in table["ks"]]
    kmax = ks[-1]

    def put(arr, dt, name):
        np.ascontiguousarray(arr, dtype=dt).tofile(os.path.join(path, name))
    put(table["pair_hash"], "<u8", "k%d.pair_hash.u64" % kmax)
    put(table["pair_gen"], "<u4", "k%d.pair_gen.u32" % kmax)
    put(table["gsize"], "<u4", "k%d.gsize.u32" % kmax)
    if table.get("kmer_hi") is not None:
        put(table["kmer_hi"], "<u8", "k%d.kmer_hi.u64" % kmax)
        put(table["kmer_lo"], "<u8", "k%d.kmer_lo.u64" % kmax)
    if filter_bits is not None:
        put(filter_bits, "<u4", "k%d.filter.u32" % kmax)
    for k in ks[:-1]:
        t = table["small"][k]
        put(t["pa"], "<u4", "k%d.rp_pa.u32" % k)
        put(t["pb"], "<u4", "k%d.rp_pb.u32" % k)
        put(t["cid"], "<u4", "k%d.rp_cid.u32" % k)
        put(t["cgen"], "<u4", "k%d.rp_cgen.u32" % k)
        put(t["gsize"], "<u4", "k%d.gsize.u32" % k)
    meta = {"format": TABLE_FORMAT, "version": 3, "n": int(n), "ks": ks, "ngenomes": len(names), "hash_mode": int(hash_mode),
            "stage_a_definition": "reference_pipeline", "sketch_hash": str(sketch_hash), "nprefix": {str(k): int(table["small"][k]["nprefix"]) for k in ks[:-1]},
            "layout": "hash-major pairs of the largest k + prefix numbers and count lists of the smaller k",
            "hash": "murmur3_x64_128.h1(canonical ASCII k-mer), seed 0" if not hash_mode else
                    "min(murmur3_x64_128.h1(k-mer), murmur3_x64_128.h1(reverse complement)) % 9999999999971"}
    with open(os.path.join(path, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


def write_sketch_table_v1(path, names, ks, n, per_k):
    """The genome-major layout of version 1 (kept for tables already on disk and for the tests of the reader)."""
    _write_common(path, names)
    for k in ks:
        h, o = per_k[k]
        np.ascontiguousarray(h, dtype="<u8").tofile(os.path.join(path, "k%d.hashes.u64" % k))
        np.ascontiguousarray(o, dtype="<u8").tofile(os.path.join(path, "k%d.offsets.u64" % k))
    meta = {"format": TABLE_FORMAT, "version": 1, "n": int(n), "ks": [int(k) for k in ks], "ngenomes": len(names),
            "hash": "murmur3_x64_128.h1(canonical ASCII k-mer), seed 0"}
    with open(os.path.join(path, "meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1)


def default_table_dir(data_dir):
    return os.path.join(data_dir, "sketch_table")


def is_gzip(path):
    with open(path, 'rb') as fh:
        return fh.read(2) == b'\x1f\x8b'


def inflate_file(path, block=16 << 20):
    """A gzip file's text (every member of it: bgzip, `cat a.gz b.gz`), inflated with zlib on large blocks — the gzip
    module's file object spends most of its time in small reads.  OSError for a stream that ends inside a member."""
    import zlib
    out = []
    with open(path, 'rb') as fh:
        d, fed = zlib.decompressobj(47), False
        while True:
            buf = fh.read(block)
            if not buf:
                break
            while buf:
                fed = True
                out.append(d.decompress(buf))
                if d.eof:  # a member ended: another may follow
                    buf, d, fed = d.unused_data, zlib.decompressobj(47), False
                    if buf and not buf.startswith(b'\x1f\x8b'[:len(buf)]):
                        return b''.join(out)  # trailing garbage (zero padding, a tape block's fill): ignored, as gzip / zcat do
                else:
                    buf = b''
        if fed:
            raise OSError('%s: the gzip stream ends in the middle of a member' % path)
    return b''.join(out)
