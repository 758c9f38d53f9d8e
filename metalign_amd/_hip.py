"""ctypes binding of libmetalign_hip.so (include/metalign_hip.h).

The HIP library is the ONLY compute path of this package: if the shared object
is missing or no MI355X is visible, every entry point raises
`HipUnavailable` — there is no CPU fallback (the CPU oracle under oracle/ is
test infrastructure and is never imported from here).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MG_LIB_PATH") or os.path.join(_HERE, "libmetalign_hip.so")  # (MG_LIB_PATH: instrumented builds of the same library)

REC_DTYPE = np.dtype([("ref_new", "<u4"), ("matched", "<u4"), ("total", "<u4"), ("flag_len", "<u4")])
NEW_BIT = 0x80000000
REF_MASK = 0x7FFFFFFF
LEN_SHIFT = 12
MAX_SEQLEN = (1 << 20) - 1
U64_MAX = 0xFFFFFFFFFFFFFFFF
MAX_K = 64

# every symbol include/metalign_hip.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "mg_abi_version", "mg_debug_set", "mg_debug_get", "mg_device_count", "mg_init", "mg_init_on_stream", "mg_shutdown", "mg_last_error",
    "mg_device_name", "mg_mem_info", "mg_mem_trim", "mg_dev_malloc", "mg_dev_free", "mg_memcpy_h2d", "mg_memcpy_d2h", "mg_dev_memset", "mg_sync",
    "mg_host_alloc", "mg_host_free", "mg_memcpy_d2h_async", "mg_memcpy_h2d_async",
    "mg_event_create", "mg_event_record", "mg_event_synchronize", "mg_event_destroy", "mg_stage_c_side_stream", "mg_stage_a_side_stream", "mg_stage_a_workgroups_per_cu", "mg_stage_c_join",
    "mg_prof_enable", "mg_prof_only", "mg_prof_reset", "mg_prof_get",
    "mg_refdb_index_kmers", "mg_refdb_has_kmer_index", "mg_refdb_distinct_kmers", "mg_refdb_kmer_heads", "mg_kcounts_new", "mg_kcounts_reset", "mg_count_kmers_dev",
    "mg_kcounts_stats", "mg_kcounts_download", "mg_kcounts_device", "mg_kcounts_wait", "mg_kcounts_pack2_bytes", "mg_kcounts_pack2_dev", "mg_kcounts_merge2_dev", "mg_kcounts_free", "mg_refpipe_mark_counts_dev", "mg_refpipe_mark_counts_ptr_dev",
    "mg_refpipe_containment_counts_dev", "mg_refdb_set_count_share",
    "mg_set_count_saturation", "mg_count_saturation", "mg_set_hash_mode", "mg_hash_mode", "mg_hash_mode1_ks", "mg_sketch_reads_dev", "mg_sketch_reads_dev_async", "mg_sketch_reads_multi_dev_async", "mg_sketch_resolve", "mg_filter_build", "mg_filter_download", "mg_filter_from_bits", "mg_filter_log2_bits", "mg_filter_make_resident", "mg_filter_drop_resident", "mg_filter_use_resident", "mg_filter_resident_bytes", "mg_filter_free",
    "mg_sketch_reads_filtered_dev", "mg_sketch_reads_filtered_dev_async", "mg_sketch_from_pairs_dev", "mg_sketch_merge_dev", "mg_sketch_merge_dev_async", "mg_sketch_split", "mg_sketch_slice_words_dev", "mg_sketch_set_bound", "mg_sketch_size", "mg_sketch_truncated", "mg_sketch_last_hash",
    "mg_sketch_kmers_seen", "mg_sketch_device_ptrs", "mg_sketch_download", "mg_sketch_free", "mg_sketch_reads",
    "mg_sketch_stream_begin", "mg_sketch_stream_begin_counts", "mg_sketch_stream_add_dev", "mg_sketch_stream_add_file", "mg_sketch_stream_finish", "mg_sketch_stream_nreads", "mg_sketch_stream_nbases", "mg_sketch_stream_free",
    "mg_reads_parse_dev", "mg_reads_parse_prefix_dev", "mg_reads_parse", "mg_reads_count", "mg_reads_nbases", "mg_reads_device_ptrs",
    "mg_reads_download", "mg_reads_free",
    "mg_acc_index_build", "mg_acc_index_free", "mg_sam_tokenize_dev", "mg_sam_tokenize", "mg_paf_tokenize_dev", "mg_paf_tokenize", "mg_sam_stream_file", "mg_sam_batch_count",
    "mg_sam_batch_last_qname", "mg_sam_batch_device_ptr", "mg_sam_batch_download", "mg_sam_batch_free",
    "mg_gunzip_open", "mg_gunzip_read", "mg_gunzip_close", "mg_zcat_files", "mg_stream_thin_file",
    "mg_inflate_dev", "mg_inflated_bytes", "mg_inflated_download", "mg_inflated_free", "mg_inflate_config", "mg_inflate_stats",
    "mg_sketch_genomes", "mg_sketch_genomes_prefix", "mg_db_upload", "mg_db_upload_sorted", "mg_db_ngenomes", "mg_db_max_hash", "mg_db_free",
    "mg_containment_dev", "mg_containment_multi_dev", "mg_containment",
    "mg_sketch_genomes_kmers", "mg_sketch_genomes_kmers_forward", "mg_refdb_build", "mg_refdb_upload", "mg_refdb_upload_begin", "mg_refdb_sizes", "mg_refdb_download_kmax", "mg_refdb_download_k", "mg_refdb_nk",
    "mg_refdb_ngenomes", "mg_refdb_max_hash", "mg_refdb_kmax_table", "mg_refdb_free", "mg_refpipe_containment_dev", "mg_refpipe_mark_dev",
    "mg_refpipe_count_dev", "mg_refdb_marks",
    "mg_profile_begin_dev", "mg_profile_acc_reset", "mg_profile_map_launch", "mg_profile_state_map", "mg_profile_map_words_dev", "mg_profile_ngroups", "mg_profile_commit_dev", "mg_profile_commit_reset_dev",
    "mg_profile_multimapped_size", "mg_profile_multimapped", "mg_profile_resolve_multimapped_dev", "mg_multimapped_shares", "mg_profile_free", "mg_profile_assign",
]


class HipUnavailable(RuntimeError):
    pass


class HipError(RuntimeError):
    """A failed library call; `code` is the mg_status value (include/metalign_hip.h), None when raised on the host side."""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code


ERR_HIP, ERR_ARG, ERR_CAPACITY, ERR_STATE, ERR_NOMEM = -1, -2, -3, -4, -5


_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p


def _np(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def load_library(path=LIB_PATH):
    """dlopen the C-ABI library and declare return types; raises HipUnavailable when absent."""
    if not os.path.exists(path):
        raise HipUnavailable(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    lib.mg_last_error.restype = ctypes.c_char_p
    for name in ("mg_reads_count", "mg_reads_nbases", "mg_sam_batch_count", "mg_sketch_stream_nreads", "mg_sketch_stream_nbases"):
        getattr(lib, name).restype = ctypes.c_uint64
    lib.mg_sam_batch_last_qname.restype = ctypes.c_char_p
    for name in ("mg_reads_free", "mg_acc_index_free", "mg_sam_batch_free"):
        getattr(lib, name).restype = None
    for name in ("mg_sketch_size", "mg_sketch_kmers_seen", "mg_sketch_last_hash", "mg_db_ngenomes", "mg_db_max_hash", "mg_profile_ngroups"):
        getattr(lib, name).restype = ctypes.c_uint64
    lib.mg_sketch_free.restype = None
    lib.mg_sketch_stream_free.restype = None
    lib.mg_filter_free.restype = None
    lib.mg_filter_log2_bits.restype = ctypes.c_uint
    lib.mg_filter_resident_bytes.restype = ctypes.c_uint64
    lib.mg_count_saturation.restype = ctypes.c_uint32
    lib.mg_db_free.restype = None
    lib.mg_refdb_free.restype = None
    lib.mg_gunzip_close.restype = None
    lib.mg_debug_get.restype = ctypes.c_int64
    lib.mg_inflated_bytes.restype = ctypes.c_uint64
    lib.mg_inflated_free.restype = None
    lib.mg_refdb_kmax_table.restype = ctypes.c_void_p
    lib.mg_refdb_ngenomes.restype = ctypes.c_uint64
    lib.mg_refdb_max_hash.restype = ctypes.c_uint64
    lib.mg_refdb_distinct_kmers.restype = ctypes.c_uint64
    lib.mg_kcounts_free.restype = None
    lib.mg_profile_free.restype = None
    lib.mg_shutdown.restype = None
    return lib


_READS_FORMAT = {"fastq": 0, "fasta": 1, "fasta_ml": 2}

_host_lib = None


def debug_set(key=None, value=0):
    """mg_debug_set: a test / diagnostic knob of the library (include/metalign_hip.h lists them); key = None: all back to their
    defaults.  The library reads no environment variable.  Needs no device."""
    global _host_lib
    if _host_lib is None:
        _host_lib = load_library()
    rc = _host_lib.mg_debug_set(key.encode() if key is not None else None, ctypes.c_int64(int(value)))
    if rc != 0:
        raise HipError("libmetalign_hip rc=%d: %s" % (rc, _host_lib.mg_last_error().decode("utf-8", "replace")), rc)


def debug_get(key):
    global _host_lib
    if _host_lib is None:
        _host_lib = load_library()
    return int(_host_lib.mg_debug_get(key.encode()))


def multimapped_shares(mm_offsets, mm_tax, mm_hitlen, weight, genome_len=None):
    """mg_multimapped_shares: resolve_multi_prop's additions per taxon from the multimapped CSR, in the reference's order of
    additions (host code of the library: no device involved).  -> (extra float64[ntax], touched bool[ntax])."""
    global _host_lib
    if _host_lib is None:
        _host_lib = load_library()
    off = np.ascontiguousarray(mm_offsets, dtype=np.uint64)
    tax = np.ascontiguousarray(mm_tax, dtype=np.uint32)
    hl = np.ascontiguousarray(mm_hitlen, dtype=np.uint64)
    w = np.ascontiguousarray(weight, dtype=np.float64)
    gl = np.ascontiguousarray(genome_len, dtype=np.float64) if genome_len is not None else None
    extra = np.zeros(len(w), dtype=np.float64)
    touched = np.zeros(len(w), dtype=np.uint8)
    one = np.zeros(1, dtype=np.uint64)
    rc = _host_lib.mg_multimapped_shares(_np(off if off.size else one, ctypes.c_uint64), ctypes.c_uint64(len(hl)),
                                         _np(tax if tax.size else np.zeros(1, np.uint32), ctypes.c_uint32),
                                         _np(hl if hl.size else one, ctypes.c_uint64), _np(w, ctypes.c_double),
                                         ctypes.c_uint32(len(w)), _np(gl, ctypes.c_double) if gl is not None else None,
                                         _np(extra, ctypes.c_double), _np(touched, ctypes.c_uint8))
    if rc != 0:
        raise HipError("libmetalign_hip rc=%d: %s" % (rc, _host_lib.mg_last_error().decode("utf-8", "replace")), rc)
    return extra, touched.astype(bool)


def hash_mode1_ks():
    """The k hash mode 1 (the CMash recollection) is built for.  Host code of the library: no device involved."""
    global _host_lib
    if _host_lib is None:
        _host_lib = load_library()
    _host_lib.mg_hash_mode1_ks.restype = ctypes.c_char_p
    return [int(x) for x in _host_lib.mg_hash_mode1_ks().decode().split(",")]


def gunzip_file(path, nthreads=0, piece=256 << 20):
    """A gzip file's text (every member; trailing garbage ignored) through the library's parallel inflater (mg_gunzip_*: one
    stream entered in the middle by many host threads).  Host code of the library: no device involved.  OSError for a corrupt
    or truncated stream."""
    global _host_lib
    if _host_lib is None:
        _host_lib = load_library()
    h = _vp()
    rc = _host_lib.mg_gunzip_open(path.encode(), ctypes.c_int(nthreads), ctypes.byref(h))
    if rc != 0:
        raise OSError("%s: %s" % (path, _host_lib.mg_last_error().decode("utf-8", "replace")))
    out = []
    try:
        while True:
            buf = np.empty(piece, dtype=np.uint8)
            n = ctypes.c_uint64(0)
            rc = _host_lib.mg_gunzip_read(h, _np(buf, ctypes.c_uint8), ctypes.c_uint64(piece), ctypes.byref(n))
            if rc != 0:
                raise OSError("%s: %s" % (path, _host_lib.mg_last_error().decode("utf-8", "replace")))
            if n.value:
                out.append(buf[: n.value])
            if n.value < piece:
                break
    finally:
        _host_lib.mg_gunzip_close(h)
    return out[0].tobytes() if len(out) == 1 else b"".join(x.tobytes() for x in out)


def zcat_files(paths, out_path, nthreads=0):
    """mg_zcat_files: every file's inflated text (all members), in order, into out_path (created / truncated); a file that is not
    gzip, corrupt or truncated contributes nothing and a `zcat:` line on stderr.  Host code of the library: no device involved.
    -> (bytes written, bool[len(paths)] failed)."""
    global _host_lib
    if _host_lib is None:
        _host_lib = load_library()
    n = len(paths)
    arr = (ctypes.c_char_p * max(n, 1))(*[os.fsencode(p) for p in paths])
    failed = np.zeros(max(n, 1), dtype=np.uint8)
    nbytes = ctypes.c_uint64(0)
    rc = _host_lib.mg_zcat_files(arr, ctypes.c_uint64(n), os.fsencode(out_path), ctypes.c_int(nthreads), ctypes.byref(nbytes),
                                 _np(failed, ctypes.c_uint8))
    if rc != 0:
        raise OSError(_host_lib.mg_last_error().decode("utf-8", "replace"))
    return nbytes.value, failed[:n].astype(bool)


def thin_file(path, kind, out_path, piece_bytes=0, nthreads=4):
    """mg_stream_thin_file (diagnostic, host only): what the streaming readers hand to the device for a plain "fastq" / "sam" file,
    written to out_path.  HipError with the stream's own codes (ERR_ARG: malformed; ERR_CAPACITY: a record does not fit a piece)."""
    global _host_lib
    if _host_lib is None:
        _host_lib = load_library()
    rc = _host_lib.mg_stream_thin_file(os.fsencode(path), ctypes.c_int({"fastq": 0, "sam": 1}[kind]), ctypes.c_uint64(int(piece_bytes)),
                                       ctypes.c_int(int(nthreads)), os.fsencode(out_path))
    if rc != 0:
        raise HipError("libmetalign_hip rc=%d: %s" % (rc, _host_lib.mg_last_error().decode("utf-8", "replace")), rc)


class DeviceArray:
    """A caller-owned HBM allocation (mg_dev_malloc) with a numpy-like shape/dtype."""

    def __init__(self, hip, count, dtype):
        self.hip = hip
        self.dtype = np.dtype(dtype)
        self.count = int(count)
        self.nbytes = self.count * self.dtype.itemsize
        p = _vp()
        hip._chk(hip.lib.mg_dev_malloc(ctypes.byref(p), ctypes.c_uint64(self.nbytes + 16)))
        self.ptr = p.value

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=self.dtype)
        assert host.size == self.count, (host.size, self.count)
        self.hip._chk(self.hip.lib.mg_memcpy_h2d(_vp(self.ptr), _vp(host.ctypes.data), ctypes.c_uint64(self.nbytes)))
        return self

    def download(self):
        out = np.empty(self.count, dtype=self.dtype)
        self.hip._chk(self.hip.lib.mg_memcpy_d2h(_vp(out.ctypes.data), _vp(self.ptr), ctypes.c_uint64(self.nbytes)))
        return out

    def memset(self, byte_value=0):
        self.hip._chk(self.hip.lib.mg_dev_memset(_vp(self.ptr), ctypes.c_int(byte_value), ctypes.c_uint64(self.nbytes)))
        return self

    def free(self):
        if self.ptr:
            self.hip.lib.mg_dev_free(_vp(self.ptr))
            self.ptr = None

    def __del__(self):  # best effort
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class PinnedArray:
    """Page-locked host memory (mg_host_alloc) viewed as a numpy array: the target of asynchronous read-backs."""

    def __init__(self, hip, count, dtype):
        self.hip = hip
        self.dtype = np.dtype(dtype)
        self.count = int(count)
        self.nbytes = self.count * self.dtype.itemsize
        p = _vp()
        hip._chk(hip.lib.mg_host_alloc(ctypes.byref(p), ctypes.c_uint64(self.nbytes + 16)))
        self.ptr = p.value
        buf = (ctypes.c_char * max(self.nbytes, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=self.dtype, count=self.count)

    def fetch_async(self, d_ptr, nbytes=None):
        """Queue device -> this buffer on the library stream; valid after hip.sync()."""
        n = self.nbytes if nbytes is None else int(nbytes)
        self.hip._chk(self.hip.lib.mg_memcpy_d2h_async(_vp(self.ptr), _vp(d_ptr), ctypes.c_uint64(n)))

    def push_async(self, d_ptr, nbytes=None):
        """Queue this buffer -> device on the library stream; do not rewrite it before that has run."""
        n = self.nbytes if nbytes is None else int(nbytes)
        self.hip._chk(self.hip.lib.mg_memcpy_h2d_async(_vp(d_ptr), _vp(self.ptr), ctypes.c_uint64(n)))

    def free(self):
        if self.ptr:
            self.array = None
            self.hip.lib.mg_host_free(_vp(self.ptr))
            self.ptr = None

    def __del__(self):  # best effort
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class Event:
    """A marker on the library's main stream (mg_event_*)."""

    def __init__(self, hip):
        self.hip = hip
        p = _vp()
        hip._chk(hip.lib.mg_event_create(ctypes.byref(p)))
        self.handle = p

    def record(self):
        self.hip._chk(self.hip.lib.mg_event_record(self.handle))

    def synchronize(self):
        self.hip._chk(self.hip.lib.mg_event_synchronize(self.handle))

    def free(self):
        if self.handle:
            self.hip.lib.mg_event_destroy(self.handle)
            self.handle = None

    def __del__(self):  # best effort
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class Sketch:
    """Device-resident read sketch for one k (opaque mg_sketch handle)."""

    def __init__(self, hip, handle, k):
        self.hip, self.handle, self.k = hip, handle, k

    @property
    def size(self):
        return int(self.hip.lib.mg_sketch_size(self.handle))

    @property
    def truncated(self):
        return bool(self.hip.lib.mg_sketch_truncated(self.handle))

    @property
    def last_hash(self):
        return int(self.hip.lib.mg_sketch_last_hash(self.handle))

    @property
    def kmers_seen(self):
        return int(self.hip.lib.mg_sketch_kmers_seen(self.handle))

    def resolve(self):
        """Finalise a deferred sketch; True when it had to be rebuilt (results derived from it are stale)."""
        rebuilt = ctypes.c_int(0)
        self.hip._chk(self.hip.lib.mg_sketch_resolve(self.handle, ctypes.byref(rebuilt)))
        return bool(rebuilt.value)

    def split(self, bounds):
        """Number of entries below each hash bound (ascending python ints)."""
        b = np.asarray([int(x) for x in bounds], dtype=np.uint64)
        out = np.zeros(len(b), dtype=np.uint64)
        if len(b):
            self.hip._chk(self.hip.lib.mg_sketch_split(self.handle, _np(b, ctypes.c_uint64), ctypes.c_uint32(len(b)),
                                                       _np(out, ctypes.c_uint64)))
        return [int(x) for x in out]

    def slice_words_dev(self, d_bounds, nbounds, d_out):
        """Slice sizes + (truncated, last hash, n, overflows) as int64 words on the device; no synchronisation."""
        self.hip._chk(self.hip.lib.mg_sketch_slice_words_dev(self.handle, _vp(d_bounds), ctypes.c_uint32(int(nbounds)),
                                                             _vp(d_out)))

    def set_bound(self, truncated, bound):
        self.hip._chk(self.hip.lib.mg_sketch_set_bound(self.handle, ctypes.c_int(int(truncated)),
                                                       ctypes.c_uint64(int(bound))))

    def device_ptrs(self):
        h, c = _vp(), _vp()
        self.hip._chk(self.hip.lib.mg_sketch_device_ptrs(self.handle, ctypes.byref(h), ctypes.byref(c)))
        return h.value, c.value

    def download(self):
        n = self.size
        h = np.empty(n, dtype=np.uint64)
        c = np.empty(n, dtype=np.uint32)
        self.hip._chk(self.hip.lib.mg_sketch_download(self.handle, _np(h, ctypes.c_uint64), _np(c, ctypes.c_uint32),
                                                      ctypes.c_uint64(n)))
        return h, c

    def free(self):
        if self.handle:
            self.hip.lib.mg_sketch_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class SamParseError(Exception):
    """A SAM line the reference cannot parse; `kind` follows include/metalign_hip.h, `line` is 0-based in the chunk."""

    def __init__(self, kind, line):
        super().__init__("SAM parse error kind %d at line %d" % (kind, line))
        self.kind, self.line = kind, line


class SketchStream:
    """mg_sketch_stream_*: every piece of a sample is hashed into the SAME per-k counting tables; finish() gives the
    sketches (pending, like sketch_reads_multi_dev_async's) of the concatenated pieces."""

    def __init__(self, hip, ks, hmaxs, s=0, filters=None, expect_bases=0, counts=None):
        self.hip, self.ks = hip, [int(k) for k in ks]
        if counts is not None:  # the pieces are counted by k-mer identity into a KmerCounts (mg_sketch_stream_begin_counts)
            self.ks, self.filts, self.counts = [], [], counts
            h = _vp()
            hip._chk(hip.lib.mg_sketch_stream_begin_counts(counts.table.handle, counts.handle, ctypes.byref(h)))
            self.handle = h
            return
        nk = len(self.ks)
        self.filts = list(filters) if filters is not None else [None] * nk
        c_ks = (ctypes.c_int * nk)(*self.ks)
        c_hm = (ctypes.c_uint64 * nk)(*[int(h) for h in hmaxs])
        c_f = (_vp * nk)(*[(f.handle if f is not None else None) for f in self.filts])
        h = _vp()
        hip._chk(hip.lib.mg_sketch_stream_begin(ctypes.c_int(nk), c_ks, c_hm, ctypes.c_uint64(int(s)), c_f,
                                                ctypes.c_uint64(int(expect_bases)), ctypes.byref(h)))
        self.handle = h

    def add_dev(self, d_bases, d_offsets, nreads, nbases=0):
        self.hip._chk(self.hip.lib.mg_sketch_stream_add_dev(self.handle, _vp(d_bases), _vp(d_offsets), ctypes.c_uint64(int(nreads)),
                                                            ctypes.c_uint64(int(nbases))))

    def add_reads(self, reads):
        d_b, d_o = reads.device_ptrs()
        self.add_dev(d_b, d_o, reads.count, getattr(reads, "nbases", 0))

    def add_file(self, path, fmt, offset=0, length=0, chunk_bytes=0, nthreads=0):
        """The file (plain, gzip or BGZF) -> page-locked chunks -> HBM -> device parser -> the tables, pipelined inside the
        library (mg_stream.hip).  offset / length: a record-aligned byte range of a plain file."""
        self.hip._chk(self.hip.lib.mg_sketch_stream_add_file(self.handle, os.fsencode(path), ctypes.c_int(_READS_FORMAT[fmt]),
                                                             ctypes.c_uint64(int(offset)), ctypes.c_uint64(int(length)),
                                                             ctypes.c_uint64(int(chunk_bytes)), ctypes.c_int(int(nthreads))))

    @property
    def nreads(self):
        return int(self.hip.lib.mg_sketch_stream_nreads(self.handle))

    @property
    def nbases(self):
        return int(self.hip.lib.mg_sketch_stream_nbases(self.handle))

    def finish(self):
        nk = len(self.ks)
        if nk == 0:
            return []
        c_out = (_vp * nk)()
        self.hip._chk(self.hip.lib.mg_sketch_stream_finish(self.handle, c_out))
        out = []
        for i in range(nk):
            sk = Sketch(self.hip, _vp(c_out[i]), self.ks[i])
            sk.filt = self.filts[i]
            out.append(sk)
        return out

    def free(self):
        if self.handle:
            self.hip.lib.mg_sketch_stream_free(self.handle)
            self.handle = None


class Reads:
    """Device-resident reads (bases + offsets) produced by the on-device FASTQ / FASTA parser."""

    def __init__(self, hip, handle):
        self.hip, self.handle = hip, handle

    @property
    def count(self):
        return int(self.hip.lib.mg_reads_count(self.handle))

    @property
    def nbases(self):
        return int(self.hip.lib.mg_reads_nbases(self.handle))

    def device_ptrs(self):
        b, o = _vp(), _vp()
        self.hip._chk(self.hip.lib.mg_reads_device_ptrs(self.handle, ctypes.byref(b), ctypes.byref(o)))
        return b.value, o.value

    def download(self):
        bases = np.empty(max(self.nbases, 1), dtype=np.uint8)
        offs = np.empty(self.count + 1, dtype=np.uint64)
        self.hip._chk(self.hip.lib.mg_reads_download(self.handle, _np(bases, ctypes.c_uint8), _np(offs, ctypes.c_uint64)))
        return bases[: self.nbases], offs

    def free(self):
        if self.handle:
            self.hip.lib.mg_reads_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class AccIndex:
    """Device-resident accession -> row table for the SAM tokeniser."""

    def __init__(self, hip, handle):
        self.hip, self.handle = hip, handle

    def free(self):
        if self.handle:
            self.hip.lib.mg_acc_index_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class Filter:
    """Membership pre-filter over a genome table's hashes (mg_filter): bit (h mod 2^b) per hash."""

    def __init__(self, hip, handle):
        self.hip, self.handle = hip, handle

    @property
    def log2_bits(self):
        return int(self.hip.lib.mg_filter_log2_bits(self.handle))

    def download(self):
        """The bit array (uint32 words), as the table builder stores it next to the table."""
        bits = np.empty((1 << self.log2_bits) // 32, dtype=np.uint32)
        self.hip._chk(self.hip.lib.mg_filter_download(self.handle, _np(bits, ctypes.c_uint32), ctypes.c_uint64(bits.nbytes)))
        return bits

    def make_resident(self, hashes, hmax, spread=0):
        """Seed the table's RESIDENT INDEX from all its hashes (mg_filter_make_resident): sketch calls given this filter then
        count in it — one random access per candidate, no filter word, no table clear.  False when the hashes crowd some
        range or the device has no room for it (the filter stays a bit filter); raises on anything else.  spread = 1: half the load, twice the memory."""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        rc = self.hip.lib.mg_filter_make_resident(self.handle, _np(hashes if hashes.size else np.zeros(1, np.uint64), ctypes.c_uint64),
                                                  ctypes.c_uint64(hashes.size), ctypes.c_uint64(int(hmax)), ctypes.c_uint(int(spread)))
        if rc in (ERR_CAPACITY, ERR_NOMEM):  # (no room for the index either: the filter stays what it was)
            return False
        self.hip._chk(rc)
        return True

    def use_resident(self, on=True):
        """False: sketch calls take the bit filter although the index exists (mg_filter_use_resident)."""
        self.hip._chk(self.hip.lib.mg_filter_use_resident(self.handle, ctypes.c_int(1 if on else 0)))

    def drop_resident(self):
        """Back to a bit filter: the resident index is freed."""
        self.hip._chk(self.hip.lib.mg_filter_drop_resident(self.handle))

    @property
    def resident_bytes(self):
        return int(self.hip.lib.mg_filter_resident_bytes(self.handle))

    def free(self):
        if self.handle:
            self.hip.lib.mg_filter_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class SketchTable:
    """Device-resident genome sketch table for one k (opaque mg_db handle)."""

    def __init__(self, hip, handle):
        self.hip, self.handle = hip, handle

    @property
    def ngenomes(self):
        return int(self.hip.lib.mg_db_ngenomes(self.handle))

    @property
    def max_hash(self):
        return int(self.hip.lib.mg_db_max_hash(self.handle))

    def free(self):
        if self.handle:
            self.hip.lib.mg_db_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class RefTable:
    """Device-resident table of the REFERENCE PIPELINE (opaque mg_refdb handle; include/metalign_hip.h): the hash-major table
    of the largest k plus, per smaller k, the prefix numbers of every pair and the count list."""

    def __init__(self, hip, handle, ks):
        self.hip, self.handle, self.ks = hip, handle, [int(k) for k in ks]

    @property
    def ngenomes(self):
        return int(self.hip.lib.mg_refdb_ngenomes(self.handle))

    @property
    def max_hash(self):
        return int(self.hip.lib.mg_refdb_max_hash(self.handle))

    def kmax_table(self):
        """The table of the largest k as a SketchTable view (owned by this handle: do not free it)."""
        ptr = self.hip.lib.mg_refdb_kmax_table(self.handle)  # (waits for a table that is still on its way up, and checks it)
        if not ptr:
            raise HipError("libmetalign_hip: %s" % self.hip.lib.mg_last_error().decode("utf-8", "replace"), ERR_ARG)
        t = SketchTable(self.hip, _vp(ptr))
        t.free = lambda: None
        t._owner = self
        return t

    def sizes(self):
        """-> (npairs, nprefix[nk-1], ncount[nk-1])"""
        m = max(len(self.ks) - 1, 1)
        npairs, npre, nc = ctypes.c_uint64(0), (ctypes.c_uint64 * m)(), (ctypes.c_uint64 * m)()
        self.hip._chk(self.hip.lib.mg_refdb_sizes(self.handle, ctypes.byref(npairs), npre, nc))
        return int(npairs.value), [int(x) for x in npre][: len(self.ks) - 1], [int(x) for x in nc][: len(self.ks) - 1]

    def download(self, kmers=True):
        """Everything the table builder stores: dict(pair_hash, pair_gen, gsize, [kmer_hi, kmer_lo,] small={k: dict(pa, pb, cid,
        cgen, gsize, nprefix)})."""
        npairs, npre, nc = self.sizes()
        g = self.ngenomes
        ph, pg, gs = np.zeros(max(npairs, 1), np.uint64), np.zeros(max(npairs, 1), np.uint32), np.zeros(max(g, 1), np.uint32)
        khi = np.zeros(max(npairs, 1), np.uint64) if kmers else None
        klo = np.zeros(max(npairs, 1), np.uint64) if kmers else None
        self.hip._chk(self.hip.lib.mg_refdb_download_kmax(self.handle, _np(ph, ctypes.c_uint64), _np(pg, ctypes.c_uint32),
                                                          _np(gs, ctypes.c_uint32), _np(khi, ctypes.c_uint64) if kmers else None,
                                                          _np(klo, ctypes.c_uint64) if kmers else None))
        out = dict(ks=list(self.ks), ngenomes=g, pair_hash=ph[:npairs], pair_gen=pg[:npairs], gsize=gs[:g], small={})
        if kmers:
            out.update(kmer_hi=khi[:npairs], kmer_lo=klo[:npairs])
        for ki, k in enumerate(self.ks[:-1]):
            pa, pb = np.zeros(max(npairs, 1), np.uint32), np.zeros(max(npairs, 1), np.uint32)
            cid, cgen = np.zeros(max(nc[ki], 1), np.uint32), np.zeros(max(nc[ki], 1), np.uint32)
            gk = np.zeros(max(g, 1), np.uint32)
            self.hip._chk(self.hip.lib.mg_refdb_download_k(self.handle, ctypes.c_int(ki), _np(pa, ctypes.c_uint32), _np(pb, ctypes.c_uint32),
                                                       _np(cid, ctypes.c_uint32), _np(cgen, ctypes.c_uint32), _np(gk, ctypes.c_uint32)))
            out["small"][k] = dict(pa=pa[:npairs], pb=pb[:npairs], cid=cid[:nc[ki]], cgen=cgen[:nc[ki]], gsize=gk[:g], nprefix=npre[ki])
        return out

    def set_count_share(self, rank, world):
        """A rank of a multi-GPU job that holds the whole table: the count step streams only this rank's share of every smaller k's
        count list from now on (the columns of k < k_max are then this rank's part of them; the ranks' parts add up)."""
        self.hip._chk(self.hip.lib.mg_refdb_set_count_share(self.handle, ctypes.c_uint32(int(rank)), ctypes.c_uint32(int(world))))

    def index_kmers(self, kmer_hi=None, kmer_lo=None):
        """The index stage A BY K-MER IDENTITY reads (mg_refdb_index_kmers): over the table's distinct canonical k_max-mers,
        built on the device.  kmer_hi / kmer_lo: the pairs' k-mers (format 3: k<K>.kmer_hi.u64 / .kmer_lo.u64) — None for a table
        built here (refdb_build), which holds them."""
        if kmer_hi is None:
            self.hip._chk(self.hip.lib.mg_refdb_index_kmers(self.handle, None, None))
        else:
            hi = np.ascontiguousarray(kmer_hi, dtype=np.uint64)
            lo = np.ascontiguousarray(kmer_lo, dtype=np.uint64)
            one = np.zeros(1, np.uint64)
            self.hip._chk(self.hip.lib.mg_refdb_index_kmers(self.handle, _np(hi if hi.size else one, ctypes.c_uint64),
                                                            _np(lo if lo.size else one, ctypes.c_uint64)))
        return self

    @property
    def has_kmer_index(self):
        return bool(self.hip.lib.mg_refdb_has_kmer_index(self.handle))

    @property
    def distinct_kmers(self):
        return int(self.hip.lib.mg_refdb_distinct_kmers(self.handle))

    def kmer_heads(self):
        """u32[npairs]: for every pair the pair whose counter holds the occurrences of its k-mer (KmerCounts.device())."""
        npairs = self.sizes()[0]
        out = np.zeros(max(npairs, 1), np.uint32)
        self.hip._chk(self.hip.lib.mg_refdb_kmer_heads(self.handle, _np(out, ctypes.c_uint32)))
        return out[:npairs]

    def kmer_counts(self):
        """A sample's occurrence counters for this table (zeroed)."""
        h = _vp()
        self.hip._chk(self.hip.lib.mg_kcounts_new(self.handle, ctypes.byref(h)))
        return KmerCounts(self.hip, h, self)

    def marks(self, ki):
        """(device pointer, words) of the prefix bitmap of k number ki (after a mark call)."""
        p, n = _vp(), ctypes.c_uint64(0)
        self.hip._chk(self.hip.lib.mg_refdb_marks(self.handle, ctypes.c_int(ki), ctypes.byref(p), ctypes.byref(n)))
        return int(p.value or 0), int(n.value)

    def free(self):
        if self.handle:
            self.hip.lib.mg_refdb_free(self.handle)  # (joins the uploader's threads of a table still on its way)
            self.handle = None
        self._arrays_on_their_way = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class KmerCounts:
    """One sample's occurrence counters of a reference-pipeline table's k_max-mers, counted BY K-MER IDENTITY (opaque mg_kcounts
    handle; include/metalign_hip.h): what `kmc -ci2 -cs3` + `kmc_tools intersect` leave (scripts/select_db.py:50-59)."""

    def __init__(self, hip, handle, table):
        self.hip, self.handle, self.table = hip, handle, table

    def reset(self):
        self.hip._chk(self.hip.lib.mg_kcounts_reset(self.handle))

    def add_dev(self, d_bases, d_offsets, nreads, nbases=0):
        """The k_max-mers of a batch of reads resident on the device (one launch, no sync)."""
        self.hip._chk(self.hip.lib.mg_count_kmers_dev(_vp(d_bases), _vp(d_offsets), ctypes.c_uint64(nreads), ctypes.c_uint64(nbases),
                                                      self.table.handle, self.handle))

    def add_reads(self, reads):
        p_b, p_o = reads.device_ptrs()
        self.add_dev(p_b, p_o, reads.count, reads.nbases)

    def stats(self):
        """-> dict(kmers, runs, passed, matches) of everything added since the last reset (synchronises)."""
        out = np.zeros(12, np.uint64)  # (a clocks build of the library fills more words: wave-cycles / 64, tallies of the matching)
        self.hip._chk(self.hip.lib.mg_kcounts_stats(self.handle, _np(out, ctypes.c_uint64)))
        st = dict(kmers=int(out[0]), runs=int(out[1]), passed=int(out[2]), matches=int(out[3]))
        if out[6]:
            st.update(clk_drain=int(out[4]) * 64, clk_hits=int(out[5]) * 64, clk_kernel=int(out[6]) * 64, entries_looked_at=int(out[7]),
                      windows_scanned=int(out[8]), longest_lane_windows=int(out[9]), longest_lane_entries=int(out[10]), drains=int(out[11]))
        return st

    def download(self):
        """u32[npairs]: min(occurrences of the pair's k-mer, cs)."""
        npairs = self.table.sizes()[0]
        out = np.zeros(max(npairs, 1), np.uint32)
        self.hip._chk(self.hip.lib.mg_kcounts_download(self.handle, self.table.handle, _np(out, ctypes.c_uint32)))
        return out[:npairs]

    def device(self):
        """(device pointer of the raw u32 counters, their number)"""
        p, n = _vp(), ctypes.c_uint64(0)
        self.hip._chk(self.hip.lib.mg_kcounts_device(self.handle, ctypes.byref(p), ctypes.byref(n)))
        return int(p.value or 0), int(n.value)

    def wait(self):
        """The main stream waits for whatever was queued on the counters last (before device()'s pointer is read there)."""
        self.hip._chk(self.hip.lib.mg_kcounts_wait(self.handle))

    def pack2_bytes(self):
        """Bytes of the two-bit array pack2_dev writes (a multiple of four)."""
        self.hip.lib.mg_kcounts_pack2_bytes.restype = ctypes.c_uint64
        return int(self.hip.lib.mg_kcounts_pack2_bytes(self.handle))

    def pack2_dev(self, d_out):
        """min(counter, 3) of every pair in two bits -> device memory at d_out (pack2_bytes() bytes): what a rank of a multi-GPU
        job hands to the others (counters that saturate at 3 or below)."""
        self.hip._chk(self.hip.lib.mg_kcounts_pack2_dev(self.handle, ctypes.c_void_p(int(d_out))))

    def merge2_dev(self, d_all, nranks, stride_bytes):
        """The counters := the sum over the nranks two-bit arrays at d_all (stride_bytes apart, a multiple of four)."""
        self.hip._chk(self.hip.lib.mg_kcounts_merge2_dev(self.handle, ctypes.c_void_p(int(d_all)), ctypes.c_uint32(int(nranks)),
                                                         ctypes.c_uint64(int(stride_bytes) // 4)))

    def free(self):
        if self.handle:
            self.hip.lib.mg_kcounts_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class ProfileShard:
    """One contiguous shard of alignment records in flight (opaque mg_profile handle)."""

    def __init__(self, hip, handle):
        self.hip, self.handle = hip, handle

    def map_launch(self):
        """Queue the map-only pass now (no sync); state_map() / ngroups then only read two words back."""
        self.hip._chk(self.hip.lib.mg_profile_map_launch(self.handle))

    def map_words_dev(self, d_out3):
        """(map[0], map[1], reads) as int64 words on the device behind the map-only pass; no synchronisation."""
        self.hip._chk(self.hip.lib.mg_profile_map_words_dev(self.handle, _vp(d_out3)))

    def state_map(self):
        m = (ctypes.c_uint8 * 2)()
        self.hip._chk(self.hip.lib.mg_profile_state_map(self.handle, m))
        return int(m[0]), int(m[1])

    @property
    def ngroups(self):
        return int(self.hip.lib.mg_profile_ngroups(self.handle))

    def commit(self, incoming_dropped, first_shard, group_base, d_count, d_bases, d_first_seen, d_scalars, reset=False):
        """reset=True: the accumulators are reset in the same launch that prepares the pass (a batch of its own)."""
        fn = self.hip.lib.mg_profile_commit_reset_dev if reset else self.hip.lib.mg_profile_commit_dev
        self.hip._chk(fn(
            self.handle, ctypes.c_int(int(incoming_dropped)), ctypes.c_int(int(first_shard)),
            ctypes.c_uint64(group_base), _vp(d_count), _vp(d_bases), _vp(d_first_seen), _vp(d_scalars)))

    def multimapped(self):
        nr, ne = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self.hip._chk(self.hip.lib.mg_profile_multimapped_size(self.handle, ctypes.byref(nr), ctypes.byref(ne)))
        off = np.zeros(nr.value + 1, dtype=np.uint64)
        tax = np.zeros(max(ne.value, 1), dtype=np.uint32)
        hl = np.zeros(max(nr.value, 1), dtype=np.uint64)
        rd = np.zeros(max(nr.value, 1), dtype=np.uint64)
        self.hip._chk(self.hip.lib.mg_profile_multimapped(self.handle, _np(off, ctypes.c_uint64),
                                                          _np(tax, ctypes.c_uint32), _np(hl, ctypes.c_uint64),
                                                          _np(rd, ctypes.c_uint64)))
        return off, tax[: ne.value], hl[: nr.value], rd[: nr.value]

    def multimapped_size(self):
        nr, ne = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self.hip._chk(self.hip.lib.mg_profile_multimapped_size(self.handle, ctypes.byref(nr), ctypes.byref(ne)))
        return nr.value, ne.value

    def resolve_multimapped(self, weight, genome_len=None):
        """resolve_multi_prop on the device: weight[t] = unique bases or NaN; -> extra[t] (host float64 array)."""
        hip = self.hip
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        d_w = hip.array(weight)
        d_l = hip.array(np.ascontiguousarray(genome_len, dtype=np.float64)) if genome_len is not None else None
        d_x = hip.empty(len(weight), np.float64)
        try:
            hip._chk(hip.lib.mg_profile_resolve_multimapped_dev(self.handle, _vp(d_w.ptr), _vp(d_l.ptr if d_l else None),
                                                                _vp(d_x.ptr)))
            return d_x.download()
        finally:
            d_w.free()
            d_x.free()
            if d_l:
                d_l.free()

    def free(self):
        if self.handle:
            self.hip.lib.mg_profile_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001
            pass


class SamBatch:
    """Alignment records left in HBM by the device tokeniser (mg_sam_tokenize_dev)."""

    def __init__(self, hip, handle):
        self.hip, self.handle = hip, handle
        self.count = int(hip.lib.mg_sam_batch_count(handle))
        p = _vp()
        hip._chk(hip.lib.mg_sam_batch_device_ptr(handle, ctypes.byref(p)))
        self.ptr = p.value or 0

    @property
    def last_qname(self):
        """QNAME of the last retained line ('' when there is none): what the next piece of the text is tokenised after."""
        return self.hip.lib.mg_sam_batch_last_qname(self.handle).decode("utf-8", "replace")

    def free(self):
        if self.handle:
            self.hip.lib.mg_sam_batch_free(self.handle)
            self.handle = None


class ResidentProfile:
    """A committed stage-C shard kept on the device together with the buffers it reads (multimapped CSR included)."""

    def __init__(self, shard, buffers):
        self.shard, self.buffers = shard, buffers

    def resolve_multimapped(self, weight, genome_len=None):
        return self.shard.resolve_multimapped(weight, genome_len)

    def free(self):
        if self.shard is not None:
            self.shard.free()
            for b in self.buffers:
                b.free()
            self.shard, self.buffers = None, []


class Hip:
    """Process-wide handle on the library + one device."""

    _instance = None

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        if self.lib.mg_device_count() <= 0:
            raise HipUnavailable("libmetalign_hip.so loaded but no HIP device is visible; there is no CPU fallback")
        if stream is None:
            self._chk(self.lib.mg_init(ctypes.c_int(device)))
        else:
            self._chk(self.lib.mg_init_on_stream(ctypes.c_int(device), _vp(stream)))
        self.device = device

    @classmethod
    def get(cls, device=None, stream=None):
        """The process-wide instance.  device / stream = None: whatever the live instance is bound to (device 0 and
        a stream of the library's own for the first call).  Asking for a device or a main stream OTHER than the live
        instance's raises: torch / RCCL collectives only order against the stream the library was initialised on
        (distributed.py), and a silently ignored argument would let them race with the library's kernels."""
        if cls._instance is None:
            cls._instance = cls(0 if device is None else device, stream)
            cls._instance.main_stream = stream
            return cls._instance
        inst = cls._instance
        if device is not None and int(device) != inst.device:
            raise HipError("libmetalign_hip is bound to device %d; Hip.get(device=%d) asked for another "
                           "(Hip.reset() first)" % (inst.device, int(device)))
        if stream is not None and stream != inst.main_stream:
            raise HipError("libmetalign_hip's main stream is %r; Hip.get(stream=%r) asked for another (Hip.reset() "
                           "first): collectives would not be ordered against the library's kernels"
                           % (inst.main_stream, stream))
        return inst

    @classmethod
    def reset(cls):
        if cls._instance is not None:
            for b in getattr(cls._instance, "_upload_bufs", None) or []:
                b.free()
            cls._instance._upload_bufs = None
            cls._instance.lib.mg_shutdown()
            cls._instance = None

    def _chk(self, rc):
        if rc != 0:
            raise HipError("libmetalign_hip rc=%d: %s" % (rc, self.lib.mg_last_error().decode("utf-8", "replace")), rc)

    # ---- misc ----
    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        self._chk(self.lib.mg_device_name(buf, 256))
        return buf.value.decode()

    def mem_info(self):
        """-> (free, total, pooled) bytes: the device's memory as the runtime reports it, and what the library's caching
        allocator holds for reuse."""
        f, t, p = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        self._chk(self.lib.mg_mem_info(ctypes.byref(f), ctypes.byref(t), ctypes.byref(p)))
        return f.value, t.value, p.value

    def mem_trim(self):
        """Cached (free) blocks of the library's allocator back to the runtime."""
        self._chk(self.lib.mg_mem_trim())

    def sync(self):
        self._chk(self.lib.mg_sync())

    def array(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype)
        return DeviceArray(self, host.size, host.dtype).upload(host)

    def upload_file(self, path, chunk=32 << 20, offset=0, length=None):
        """A file's bytes -> HBM through page-locked chunks: reader threads fill chunks (page cache -> pinned buffer, one
        copy, positional reads, the GIL released) while the DMA of earlier chunks runs; this thread alone talks to the
        library.  Two chunks for files up to 256 MB, four above (one thread copies ~14 GB/s out of the page cache,
        PCIe takes ~50).  The chunks stay with the instance: page-locking 32 MB costs ~5 ms each time.
        offset / length: a byte range of the file (a rank's share of a multi-GPU launch).
        -> (DeviceArray of uint8, size)."""
        from concurrent.futures import ThreadPoolExecutor
        size = os.path.getsize(path) - offset if length is None else int(length)
        dev = self.empty(max(size, 1), np.uint8)
        if size == 0:
            return dev, 0
        kept = getattr(self, "_upload_bufs", None) or []
        if kept and kept[0].count < min(chunk, size):  # kept for smaller files: start over
            for b in kept:
                b.free()
            kept = []
        bufsize = kept[0].count if kept else int(chunk if size > chunk // 4 else size)  # (every kept chunk has this size)
        chunk = int(min(bufsize, size))
        nbufs = 2 if size <= 8 * chunk else 4
        while len(kept) < nbufs:
            kept.append(self.pinned(bufsize, np.uint8))
        self._upload_bufs = bufs = kept
        nchunks = (size + chunk - 1) // chunk
        evs = [self.event() for _ in range(nbufs)]
        fd = os.open(path, os.O_RDONLY)

        def fill(i):
            off, want = i * chunk, min(chunk, size - i * chunk)
            view, got = memoryview(bufs[i % nbufs].array)[:want], 0
            while got < want:
                n = os.preadv(fd, [view[got:]], offset + off + got)
                if not n:
                    raise HipError("%s: short read at byte %d of %d" % (path, offset + off + got, offset + size))
                got += n
            return want

        try:
            with ThreadPoolExecutor(nbufs) as ex:
                futs = {i: ex.submit(fill, i) for i in range(min(nbufs, nchunks))}
                for i in range(nchunks):
                    want = futs.pop(i).result()
                    bufs[i % nbufs].push_async(dev.ptr + i * chunk, want)
                    evs[i % nbufs].record()
                    if i + nbufs < nchunks:
                        evs[i % nbufs].synchronize()  # this buffer's DMA has run: the next read may overwrite it
                        futs[i + nbufs] = ex.submit(fill, i + nbufs)
            self.sync()
        finally:
            os.close(fd)
            for ev in evs:
                ev.free()
        return dev, size

    def stage_c_side_stream(self, on=True):
        self._chk(self.lib.mg_stage_c_side_stream(ctypes.c_int(int(on))))

    def stage_a_side_stream(self, on=True):
        """0 / False: main stream (waits for the stage-A streams); 1 / True or 2: which stage-A stream is next; 3 or 4: the same two
        at the device's lowest stream priority (a single shard's passes: 5 % faster; with collectives in the path: slower)."""
        self._chk(self.lib.mg_stage_a_side_stream(ctypes.c_int(int(on))))

    def stage_a_workgroups_per_cu(self, n):
        self._chk(self.lib.mg_stage_a_workgroups_per_cu(ctypes.c_int(int(n))))

    def stage_c_join(self):
        self._chk(self.lib.mg_stage_c_join())

    def event(self):
        return Event(self)

    def pinned(self, count, dtype):
        return PinnedArray(self, count, dtype)

    def empty(self, count, dtype):
        return DeviceArray(self, count, dtype)

    def prof_enable(self, on=True):
        self._chk(self.lib.mg_prof_enable(ctypes.c_int(1 if on else 0)))

    def prof_only(self, kernel=""):
        self._chk(self.lib.mg_prof_only(kernel.encode()))

    def prof_reset(self):
        self._chk(self.lib.mg_prof_reset())

    def prof_get(self, kernel):
        n, ms = ctypes.c_uint64(0), ctypes.c_double(0.0)
        self._chk(self.lib.mg_prof_get(kernel.encode(), ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    # ---- gzip / BGZF inflated on the device (mg_inflate.hip) ----
    def inflate(self, blob):
        """mg_inflate_dev + download: a whole gzip / BGZF file's bytes -> its text (every member; trailing garbage ignored), inflated
        by the device.  OSError (zlib's wording) for a corrupt or truncated stream."""
        src = np.frombuffer(blob, dtype=np.uint8) if not isinstance(blob, np.ndarray) else np.ascontiguousarray(blob, dtype=np.uint8)
        h = _vp()
        rc = self.lib.mg_inflate_dev(_np(src if src.size else np.zeros(1, np.uint8), ctypes.c_uint8), ctypes.c_uint64(src.size), ctypes.byref(h))
        if rc != 0:
            msg = self.lib.mg_last_error().decode("utf-8", "replace")
            if rc == ERR_ARG:
                raise OSError(msg)
            raise HipError("libmetalign_hip rc=%d: %s" % (rc, msg), rc)
        try:
            n = self.lib.mg_inflated_bytes(h)
            out = np.empty(max(n, 1), dtype=np.uint8)
            self._chk(self.lib.mg_inflated_download(h, _np(out, ctypes.c_uint8)))
        finally:
            self.lib.mg_inflated_free(h)
        return out[:n].tobytes()

    def inflate_config(self, chunk_bytes=0, stage_bytes=0, ratio=0, on=-1, lane_jobs=-1):
        """mg_inflate_config: compressed bytes per job / per stage, symbols reserved per compressed byte, and whether `.gz` inputs of
        the streaming entry points are inflated on the device (on = 1 / 0), and from how many jobs a launch decodes a job per lane instead
        of a job per wavefront (lane_jobs; 0 = always per lane); -1 (chunk_bytes, stage_bytes, ratio: 0) leave a setting as it is;
        stage_bytes = -1: stages sized by the device (a round of jobs each, the default)."""
        self._chk(self.lib.mg_inflate_config(ctypes.c_int64(int(chunk_bytes)), ctypes.c_int64(int(stage_bytes)), ctypes.c_int(int(ratio)), ctypes.c_int(int(on)),
                                             ctypes.c_int64(int(lane_jobs))))

    def inflate_stats(self, reset=False):
        """-> dict of the device inflater's counters since the last reset."""
        class _C(ctypes.Structure):
            _fields_ = [("stages", ctypes.c_uint64), ("jobs", ctypes.c_uint64), ("redone", ctypes.c_uint64), ("find_candidates", ctypes.c_uint64),
                        ("find_steps", ctypes.c_uint64), ("blocks", ctypes.c_uint64), ("batches", ctypes.c_uint64), ("windows", ctypes.c_uint64),
                        ("symbols_out", ctypes.c_uint64), ("clk_tables", ctypes.c_uint64), ("clk_decode", ctypes.c_uint64), ("clk_emit", ctypes.c_uint64),
                        ("clk_tail", ctypes.c_uint64), ("clk_sub", ctypes.c_uint64 * 6), ("find_s", ctypes.c_double),
                        ("decode_s", ctypes.c_double), ("resolve_s", ctypes.c_double), ("stage_s", ctypes.c_double)]
        c = _C()
        self._chk(self.lib.mg_inflate_stats(ctypes.byref(c), ctypes.c_int(1 if reset else 0)))
        return {k: (list(getattr(c, k)) if k == "clk_sub" else getattr(c, k)) for k, _ in _C._fields_}

    # ---- stage A ----
    def count_saturation(self, cs=None):
        """Occurrence counters of read sketches saturate at cs (default 3 = kmc -cs3, select_db.py:50; 0 = exact).
        With an argument: set it for sketches built from now on.  -> the value in force."""
        if cs is not None:
            self._chk(self.lib.mg_set_count_saturation(ctypes.c_uint32(int(cs))))
        return int(self.lib.mg_count_saturation())

    def filter_build(self, hashes):
        """Membership pre-filter over ALL hashes of a genome table (one k)."""
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        h = _vp()
        self._chk(self.lib.mg_filter_build(_np(hashes if hashes.size else np.zeros(1, np.uint64), ctypes.c_uint64),
                                           ctypes.c_uint64(hashes.size), ctypes.byref(h)))
        return Filter(self, h)

    def filter_from_bits(self, bits):
        """A stored filter (Filter.download) back onto the device."""
        bits = np.ascontiguousarray(bits, dtype=np.uint32)
        lb = int(bits.size * 32).bit_length() - 1
        h = _vp()
        self._chk(self.lib.mg_filter_from_bits(_np(bits, ctypes.c_uint32), ctypes.c_uint(lb), ctypes.byref(h)))
        return Filter(self, h)

    def sketch_reads_dev(self, d_bases, d_offsets, nreads, k, hmax=U64_MAX, s=0, filt=None):
        """filt: a Filter (the table's): the sketch is restricted to hashes that may be in the table."""
        h = _vp()
        if filt is None:
            self._chk(self.lib.mg_sketch_reads_dev(_vp(d_bases), _vp(d_offsets), ctypes.c_uint64(nreads), ctypes.c_int(k),
                                                   ctypes.c_uint64(hmax), ctypes.c_uint64(s), ctypes.byref(h)))
        else:
            self._chk(self.lib.mg_sketch_reads_filtered_dev(_vp(d_bases), _vp(d_offsets), ctypes.c_uint64(nreads),
                                                            ctypes.c_int(k), ctypes.c_uint64(hmax), ctypes.c_uint64(s),
                                                            filt.handle, ctypes.byref(h)))
        sk = Sketch(self, h, k)
        sk.filt = filt  # keeps the filter alive as long as the sketch may still be rebuilt with it
        return sk

    def sketch_reads_dev_async(self, d_bases, d_offsets, nreads, k, hmax=U64_MAX, s=0, filt=None):
        """No host sync: the sketch finalises at its first host-side read (Sketch.resolve / size / download ...)."""
        h = _vp()
        if filt is None:
            self._chk(self.lib.mg_sketch_reads_dev_async(_vp(d_bases), _vp(d_offsets), ctypes.c_uint64(nreads),
                                                         ctypes.c_int(k), ctypes.c_uint64(hmax), ctypes.c_uint64(s),
                                                         ctypes.byref(h)))
        else:
            self._chk(self.lib.mg_sketch_reads_filtered_dev_async(_vp(d_bases), _vp(d_offsets), ctypes.c_uint64(nreads),
                                                                  ctypes.c_int(k), ctypes.c_uint64(hmax), ctypes.c_uint64(s),
                                                                  filt.handle, ctypes.byref(h)))
        sk = Sketch(self, h, k)
        sk.filt = filt
        return sk

    def sketch_reads_multi_dev_async(self, d_bases, d_offsets, nreads, ks, hmaxs, s=0, filts=None):
        """The read sketches for several k from ONE pass over the reads (the reference's query is multi-k: `30-60-10`,
        select_db.py:75): -> [Sketch per k], every one pending like sketch_reads_dev_async's."""
        filts = list(filts) if filts is not None else [None] * len(ks)
        nk = len(ks)
        c_ks = (ctypes.c_int * nk)(*[int(k) for k in ks])
        c_hm = (ctypes.c_uint64 * nk)(*[int(h) for h in hmaxs])
        c_f = (_vp * nk)(*[(f.handle if f is not None else None) for f in filts])
        c_out = (_vp * nk)()
        self._chk(self.lib.mg_sketch_reads_multi_dev_async(_vp(d_bases), _vp(d_offsets), ctypes.c_uint64(nreads),
                                                           ctypes.c_int(nk), c_ks, c_hm, ctypes.c_uint64(s), c_f, c_out))
        out = []
        for i in range(nk):
            sk = Sketch(self, _vp(c_out[i]), int(ks[i]))
            sk.filt = filts[i]  # keeps the filter alive as long as the sketch may still be rebuilt with it
            out.append(sk)
        return out

    def sketch_from_pairs_dev(self, d_hashes, d_counts, n, k, s=0, any_truncated=False, bound=U64_MAX):
        h = _vp()
        self._chk(self.lib.mg_sketch_from_pairs_dev(_vp(d_hashes), _vp(d_counts), ctypes.c_uint64(n),
                                                    ctypes.c_uint64(s), ctypes.c_int(int(any_truncated)),
                                                    ctypes.c_uint64(bound), ctypes.byref(h)))
        return Sketch(self, h, k)

    def sketch_merge_dev(self, d_hashes, d_counts, n, k, range_lo, range_hi, s=0, any_truncated=False, bound=U64_MAX):
        h = _vp()
        self._chk(self.lib.mg_sketch_merge_dev(_vp(d_hashes), _vp(d_counts), ctypes.c_uint64(n),
                                               ctypes.c_uint64(int(range_lo)), ctypes.c_uint64(int(range_hi)),
                                               ctypes.c_uint64(s), ctypes.c_int(int(any_truncated)),
                                               ctypes.c_uint64(bound), ctypes.byref(h)))
        return Sketch(self, h, k)

    def sketch_merge_dev_async(self, d_hashes, d_counts, n, k, range_lo, range_hi, s=0, any_truncated=False, bound=U64_MAX):
        """Queued without a host sync; the inputs must outlive Sketch.resolve()."""
        h = _vp()
        self._chk(self.lib.mg_sketch_merge_dev_async(_vp(d_hashes), _vp(d_counts), ctypes.c_uint64(n),
                                                     ctypes.c_uint64(int(range_lo)), ctypes.c_uint64(int(range_hi)),
                                                     ctypes.c_uint64(s), ctypes.c_int(int(any_truncated)),
                                                     ctypes.c_uint64(bound), ctypes.byref(h)))
        return Sketch(self, h, k)

    def sketch_reads(self, bases, offsets, k, hmax=U64_MAX, s=0):
        """Host arrays in, host arrays out: (hashes, counts, truncated, kmers_seen)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nreads = len(offsets) - 1
        d_b = self.array(bases if bases.size else np.zeros(1, np.uint8))
        d_o = self.array(offsets)
        sk = self.sketch_reads_dev(d_b.ptr, d_o.ptr, nreads, k, hmax, s)
        try:
            h, c = sk.download()
            return h, c, sk.truncated, sk.kmers_seen
        finally:
            sk.free()
            d_b.free()
            d_o.free()

    # ---- ingest ----
    def parse_reads(self, text, fmt):
        """FASTQ (fmt 'fastq'), one-sequence-line FASTA ('fasta') or any FASTA ('fasta_ml') bytes -> device-resident Reads."""
        buf = np.frombuffer(text, dtype=np.uint8) if not isinstance(text, np.ndarray) else text
        h = _vp()
        src = buf if buf.size else np.zeros(1, np.uint8)
        self._chk(self.lib.mg_reads_parse(_np(src, ctypes.c_uint8), ctypes.c_uint64(buf.size),
                                          ctypes.c_int(_READS_FORMAT[fmt]), ctypes.byref(h)))
        return Reads(self, h)

    def set_hash_mode(self, mode):
        """0: hash of the canonical k-mer (default); 1: min(hash(kmer), hash(revcomp)) % 9999999999971 — CMash as SURVEY.md
        §8(c) recollects it (include/metalign_hip.h: mg_set_hash_mode)."""
        self._chk(self.lib.mg_set_hash_mode(ctypes.c_int(int(mode))))

    @property
    def hash_mode(self):
        return int(self.lib.mg_hash_mode())

    def sketch_stream(self, ks, hmaxs, s=0, filters=None, expect_bases=0):
        """A streamed read sketch: one set of counting tables for a sample that arrives in pieces (SketchStream)."""
        return SketchStream(self, ks, hmaxs, s, filters, expect_bases)

    def count_stream(self, counts):
        """The same stream of pieces (add_file / add_dev / add_reads) counted by k-mer identity into `counts` (a KmerCounts)."""
        return SketchStream(self, [], [], counts=counts)

    def parse_reads_dev(self, d_text, nbytes, fmt):
        """As parse_reads, for text already resident in HBM."""
        h = _vp()
        self._chk(self.lib.mg_reads_parse_dev(_vp(d_text), ctypes.c_uint64(nbytes),
                                              ctypes.c_int(_READS_FORMAT[fmt]), ctypes.byref(h)))
        return Reads(self, h)

    def sam_tokenize_dev(self, d_text, nbytes, acc_index, prev_qname=""):
        """As sam_tokenize, for text already resident in HBM; returns the number of records (left on the device)."""
        h = _vp()
        kind, line = ctypes.c_int(0), ctypes.c_uint64(0)
        rc = self.lib.mg_sam_tokenize_dev(_vp(d_text), ctypes.c_uint64(nbytes), acc_index.handle,
                                          ctypes.c_char_p(prev_qname.encode()), ctypes.byref(h), ctypes.byref(kind),
                                          ctypes.byref(line))
        if rc != 0 and kind.value:
            raise SamParseError(kind.value, line.value)
        self._chk(rc)
        n = int(self.lib.mg_sam_batch_count(h))
        self.lib.mg_sam_batch_free(h)
        return n

    def sam_tokenize_dev_batch(self, d_text, nbytes, acc_index, prev_qname="", paf=False):
        """SAM (or, paf=True, PAF) text resident in HBM -> SamBatch (records stay on the device).  SamParseError as
        sam_tokenize."""
        h = _vp()
        kind, line = ctypes.c_int(0), ctypes.c_uint64(0)
        fn = self.lib.mg_paf_tokenize_dev if paf else self.lib.mg_sam_tokenize_dev
        rc = fn(_vp(d_text), ctypes.c_uint64(nbytes), acc_index.handle,
                                          ctypes.c_char_p(prev_qname.encode()), ctypes.byref(h), ctypes.byref(kind),
                                          ctypes.byref(line))
        if rc != 0 and kind.value:
            raise SamParseError(kind.value, line.value)
        self._chk(rc)
        return SamBatch(self, h)

    def sam_stream_file(self, path, acc_index, paf=False, offset=0, length=0, chunk_bytes=0, nthreads=0):
        """The alignment file (SAM; paf=True: PAF; plain, gzip or BGZF) -> SamBatch, streamed through page-locked chunks
        inside the library (mg_sam_stream_file): the file read, the upload and the tokeniser overlap, and the text never
        exists as a host array.  SamParseError for a line the reference cannot parse."""
        h = _vp()
        kind, line = ctypes.c_int(0), ctypes.c_uint64(0)
        rc = self.lib.mg_sam_stream_file(os.fsencode(path), ctypes.c_int(1 if paf else 0), acc_index.handle,
                                         ctypes.c_uint64(int(offset)), ctypes.c_uint64(int(length)), ctypes.c_uint64(int(chunk_bytes)),
                                         ctypes.c_int(int(nthreads)), ctypes.byref(h), ctypes.byref(kind), ctypes.byref(line))
        if rc != 0 and kind.value:
            raise SamParseError(kind.value, line.value)
        self._chk(rc)
        return SamBatch(self, h)

    def acc_index(self, names):
        blob = "".join(names).encode()
        offs = np.zeros(len(names) + 1, dtype=np.uint64)
        if names:
            offs[1:] = np.cumsum([len(n.encode()) for n in names])
        h = _vp()
        self._chk(self.lib.mg_acc_index_build(ctypes.c_char_p(blob), _np(offs, ctypes.c_uint64),
                                              ctypes.c_uint32(len(names)), ctypes.byref(h)))
        return AccIndex(self, h)

    def sam_tokenize(self, text, acc_index, prev_qname="", paf=False):
        """One chunk of SAM (paf=True: PAF) text (ending on a line boundary) -> (records REC_DTYPE[], QNAME of its last
        retained line).  Raises SamParseError for a line the reference cannot parse."""
        buf = np.frombuffer(text, dtype=np.uint8)  # bytes or a memoryview slice: no copy
        h = _vp()
        kind, line = ctypes.c_int(0), ctypes.c_uint64(0)
        src = buf if buf.size else np.zeros(1, np.uint8)
        fn = self.lib.mg_paf_tokenize if paf else self.lib.mg_sam_tokenize
        rc = fn(_np(src, ctypes.c_uint8), ctypes.c_uint64(buf.size), acc_index.handle,
                                      ctypes.c_char_p(prev_qname.encode()), ctypes.byref(h), ctypes.byref(kind),
                                      ctypes.byref(line))
        if rc != 0 and kind.value:
            raise SamParseError(kind.value, line.value)
        self._chk(rc)
        try:
            n = int(self.lib.mg_sam_batch_count(h))
            recs = np.zeros(n, dtype=REC_DTYPE)
            if n:
                self._chk(self.lib.mg_sam_batch_download(h, _vp(recs.ctypes.data)))
            last = self.lib.mg_sam_batch_last_qname(h).decode("utf-8", "replace")
            return recs, last
        finally:
            self.lib.mg_sam_batch_free(h)

    # ---- stage A' ----
    def sketch_genomes(self, bases, offsets, k, n):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        g = len(offsets) - 1
        out_h = np.zeros(max(g * n, 1), dtype=np.uint64)
        out_o = np.zeros(g + 1, dtype=np.uint64)
        self._chk(self.lib.mg_sketch_genomes(_np(bases, ctypes.c_uint8), _np(offsets, ctypes.c_uint64),
                                             ctypes.c_uint64(g), ctypes.c_int(k), ctypes.c_uint64(n),
                                             _np(out_h, ctypes.c_uint64), _np(out_o, ctypes.c_uint64)))
        return out_h[: int(out_o[-1])].copy(), out_o

    def sketch_genomes_prefix(self, bases, offsets, kmax, k, n):
        """The k < kmax table of hash mode 1: per genome the distinct mode-1 hashes of the k-prefixes of its sketched kmax-mers
        (mg_sketch_genomes_prefix; CMash as SURVEY.md §8(c) recollects it, unverified)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        g = len(offsets) - 1
        out_h = np.zeros(max(g * n, 1), dtype=np.uint64)
        out_o = np.zeros(g + 1, dtype=np.uint64)
        self._chk(self.lib.mg_sketch_genomes_prefix(_np(bases, ctypes.c_uint8), _np(offsets, ctypes.c_uint64), ctypes.c_uint64(g),
                                                    ctypes.c_int(kmax), ctypes.c_int(k), ctypes.c_uint64(n),
                                                    _np(out_h, ctypes.c_uint64), _np(out_o, ctypes.c_uint64)))
        return out_h[: int(out_o[-1])].copy(), out_o

    def sketch_genomes_kmers(self, bases, offsets, k, n, sketch_hash="canonical"):
        """mg_sketch_genomes plus every sketch entry's k-mer as the table keeps it, 2-bit packed (the reference pipeline's
        table is built from these: refdb_build).  -> (hashes, kmer_hi, kmer_lo, offsets[G+1]).  sketch_hash = "forward"
        (mg_sketch_genomes_kmers_forward): the entries are SELECTED by MurmurHash3(k-mer as it stands) mod 9999999999971 and kept as
        they stand; `hashes` is still what they match by (not ascending within a genome then, and not necessarily distinct)."""
        if sketch_hash not in ("canonical", "forward"):
            raise ValueError("sketch_hash: 'canonical' or 'forward'")
        fn = self.lib.mg_sketch_genomes_kmers_forward if sketch_hash == "forward" else self.lib.mg_sketch_genomes_kmers
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        g = len(offsets) - 1
        out_h, out_hi, out_lo = (np.zeros(max(g * n, 1), dtype=np.uint64) for _ in range(3))
        out_o = np.zeros(g + 1, dtype=np.uint64)
        self._chk(fn(_np(bases, ctypes.c_uint8), _np(offsets, ctypes.c_uint64), ctypes.c_uint64(g), ctypes.c_int(k), ctypes.c_uint64(n),
                     _np(out_h, ctypes.c_uint64), _np(out_hi, ctypes.c_uint64), _np(out_lo, ctypes.c_uint64), _np(out_o, ctypes.c_uint64)))
        e = int(out_o[-1])
        return out_h[:e].copy(), out_hi[:e].copy(), out_lo[:e].copy(), out_o

    def refdb_build(self, hashes, kmer_hi, kmer_lo, offsets, ks):
        """The reference pipeline's table from the genome-major entries of the largest k (sketch_genomes_kmers), on the device."""
        hashes, kmer_hi, kmer_lo = (np.ascontiguousarray(a, dtype=np.uint64) for a in (hashes, kmer_hi, kmer_lo))
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        one = np.zeros(1, np.uint64)
        c_ks = (ctypes.c_int * len(ks))(*[int(k) for k in ks])
        h = _vp()
        self._chk(self.lib.mg_refdb_build(_np(hashes if hashes.size else one, ctypes.c_uint64), _np(kmer_hi if kmer_hi.size else one, ctypes.c_uint64),
                                          _np(kmer_lo if kmer_lo.size else one, ctypes.c_uint64), _np(offsets, ctypes.c_uint64),
                                          ctypes.c_uint64(len(offsets) - 1), ctypes.c_int(len(ks)), c_ks, ctypes.byref(h)))
        return RefTable(self, h, ks)

    def refdb_upload(self, ks, ngenomes, pair_hash, pair_gen, gsize, max_hash, small, wait=True):
        """The same handle from stored arrays (formats.SketchTable.refpipe_arrays), or a rank's share of them.
        small: per k below the largest, in order: dict(pa, pb, cid, cgen, gsize, nprefix).
        wait = False (mg_refdb_upload_begin): returns while the library's threads still copy the arrays up — the handle keeps the
        arrays alive; the first call that reads the table waits for them and checks them (a corrupt table is reported there)."""
        ks = [int(k) for k in ks]
        pair_hash = np.ascontiguousarray(pair_hash, dtype=np.uint64)
        pair_gen = np.ascontiguousarray(pair_gen, dtype=np.uint32)
        gsize = np.ascontiguousarray(gsize, dtype=np.uint32)
        n, m = pair_hash.size, len(ks) - 1
        assert len(small) == m
        keep = []

        def arr(a, dt):
            a = np.ascontiguousarray(a, dtype=dt)
            if a.size == 0:
                a = np.zeros(1, dtype=dt)
            keep.append(a)
            return a.ctypes.data_as(ctypes.c_void_p)
        mm = max(m, 1)
        c_pa, c_pb, c_cid, c_cgen, c_gs = ((_vp * mm)() for _ in range(5))
        c_np, c_nc = (ctypes.c_uint64 * mm)(), (ctypes.c_uint64 * mm)()
        for i, t in enumerate(small):
            c_pa[i], c_pb[i] = arr(t["pa"], np.uint32), arr(t["pb"], np.uint32)
            c_cid[i], c_cgen[i], c_gs[i] = arr(t["cid"], np.uint32), arr(t["cgen"], np.uint32), arr(t["gsize"], np.uint32)
            c_np[i], c_nc[i] = int(t["nprefix"]), len(t["cid"])
        c_ks = (ctypes.c_int * len(ks))(*ks)
        h = _vp()
        fn = self.lib.mg_refdb_upload if wait else self.lib.mg_refdb_upload_begin
        self._chk(fn(ctypes.c_uint64(int(ngenomes)), ctypes.c_int(len(ks)), c_ks, ctypes.c_uint64(n),
                     arr(pair_hash, np.uint64), arr(pair_gen, np.uint32), arr(gsize, np.uint32),
                     ctypes.c_uint64(int(max_hash)), c_pa, c_pb, c_np, c_cid, c_cgen, c_nc, c_gs, ctypes.byref(h)))
        t = RefTable(self, h, ks)
        if not wait:
            t._arrays_on_their_way = keep  # (memory maps of the stored table: the uploader's threads read them until the first use)
        return t

    def refpipe_containment_dev(self, sketch, reftable, ci, d_hits, d_sizes):
        """Stage B of the reference pipeline: sketch = the read sketch of the table's largest k; d_hits / d_sizes: one device
        pointer per k of the table."""
        nk = len(reftable.ks)
        c_h = (_vp * nk)(*[_vp(p) for p in d_hits])
        c_s = (_vp * nk)(*[_vp(p) for p in d_sizes])
        self._chk(self.lib.mg_refpipe_containment_dev(sketch.handle, reftable.handle, ctypes.c_uint32(ci), c_h, c_s))

    def refpipe_containment_counts_dev(self, counts, reftable, ci, d_hits, d_sizes):
        """Stage B of the reference pipeline from the k-mer counters of stage A by identity (KmerCounts)."""
        nk = len(reftable.ks)
        c_h = (_vp * nk)(*[_vp(p) for p in d_hits])
        c_s = (_vp * nk)(*[_vp(p) for p in d_sizes])
        self._chk(self.lib.mg_refpipe_containment_counts_dev(counts.handle, reftable.handle, ctypes.c_uint32(ci), c_h, c_s))

    def refpipe_mark_counts_dev(self, counts, reftable, ci, d_hits_kmax, d_sizes_kmax):
        """counts: a KmerCounts, or the device pointer of counters summed over the ranks (u32[npairs])."""
        if isinstance(counts, KmerCounts):
            self._chk(self.lib.mg_refpipe_mark_counts_dev(counts.handle, reftable.handle, ctypes.c_uint32(ci), _vp(d_hits_kmax), _vp(d_sizes_kmax)))
        else:
            self._chk(self.lib.mg_refpipe_mark_counts_ptr_dev(_vp(counts), reftable.handle, ctypes.c_uint32(ci), _vp(d_hits_kmax), _vp(d_sizes_kmax)))

    def refpipe_containment_counts(self, counts, reftable, ci=2):
        """-> (hits u32[K][G], sizes u32[K][G]), k ascending."""
        g, nk = reftable.ngenomes, len(reftable.ks)
        d = self.empty(max(2 * g * nk, 1), np.uint32)
        try:
            self.refpipe_containment_counts_dev(counts, reftable, ci, [d.ptr + 4 * (2 * ki * g) for ki in range(nk)],
                                                [d.ptr + 4 * ((2 * ki + 1) * g) for ki in range(nk)])
            a = d.download()[: 2 * g * nk].reshape(nk, 2, g)
            return a[:, 0, :].copy(), a[:, 1, :].copy()
        finally:
            d.free()

    def refpipe_mark_dev(self, sketch, reftable, ci, d_hits_kmax, d_sizes_kmax):
        self._chk(self.lib.mg_refpipe_mark_dev(sketch.handle, reftable.handle, ctypes.c_uint32(ci), _vp(d_hits_kmax), _vp(d_sizes_kmax)))

    def refpipe_count_dev(self, reftable, d_marks, d_hits, d_sizes):
        """d_marks: one device pointer per smaller k (None: the handle's own bitmaps); d_hits / d_sizes: nk - 1 pointers."""
        m = max(len(reftable.ks) - 1, 1)
        c_m = (_vp * m)(*[_vp(p) for p in d_marks]) if d_marks is not None else None
        c_h = (_vp * m)(*[_vp(p) for p in d_hits])
        c_s = (_vp * m)(*[_vp(p) for p in d_sizes])
        self._chk(self.lib.mg_refpipe_count_dev(reftable.handle, c_m, c_h, c_s))

    def refpipe_containment(self, sketch, reftable, ci=2):
        """-> (hits u32[K][G], sizes u32[K][G]), k ascending."""
        g, nk = reftable.ngenomes, len(reftable.ks)
        d = self.empty(max(2 * g * nk, 1), np.uint32)
        try:
            self.refpipe_containment_dev(sketch, reftable, ci, [d.ptr + 4 * (2 * ki * g) for ki in range(nk)],
                                         [d.ptr + 4 * ((2 * ki + 1) * g) for ki in range(nk)])
            a = d.download()[: 2 * g * nk].reshape(nk, 2, g)
            return a[:, 0, :].copy(), a[:, 1, :].copy()
        finally:
            d.free()

    def upload_table(self, hashes, offsets):
        hashes = np.ascontiguousarray(hashes, dtype=np.uint64)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        if hashes.size == 0:
            hashes = np.zeros(1, dtype=np.uint64)
        h = _vp()
        self._chk(self.lib.mg_db_upload(_np(hashes, ctypes.c_uint64), _np(offsets, ctypes.c_uint64),
                                        ctypes.c_uint64(len(offsets) - 1), ctypes.byref(h)))
        return SketchTable(self, h)

    def upload_table_sorted(self, pair_hash, pair_gen, gsize, max_hash):
        """A table (or a hash-range slice of one) that is hash-major already (formats.SketchTable.pairs): no sort."""
        pair_hash = np.ascontiguousarray(pair_hash, dtype=np.uint64)
        pair_gen = np.ascontiguousarray(pair_gen, dtype=np.uint32)
        gsize = np.ascontiguousarray(gsize, dtype=np.uint32)
        n = pair_hash.size
        h = _vp()
        ph = pair_hash if n else np.zeros(1, np.uint64)
        pg = pair_gen if n else np.zeros(1, np.uint32)
        self._chk(self.lib.mg_db_upload_sorted(_np(ph, ctypes.c_uint64), _np(pg, ctypes.c_uint32), ctypes.c_uint64(n),
                                               _np(gsize, ctypes.c_uint32), ctypes.c_uint64(gsize.size),
                                               ctypes.c_uint64(int(max_hash)), ctypes.byref(h)))
        return SketchTable(self, h)

    # ---- stage B ----
    def containment_dev(self, sketch, table, ci, d_hits, d_sizes):
        self._chk(self.lib.mg_containment_dev(sketch.handle, table.handle, ctypes.c_uint32(ci), _vp(d_hits),
                                              _vp(d_sizes)))

    def containment_multi_dev(self, sketches, tables, ci, d_hits, d_sizes):
        """Stage B of every k of a pass: one launch of each kernel for all of them (mg_containment_multi_dev)."""
        nk = len(sketches)
        c_q = (_vp * nk)(*[s.handle for s in sketches])
        c_t = (_vp * nk)(*[t.handle for t in tables])
        c_h = (_vp * nk)(*[_vp(p) for p in d_hits])
        c_s = (_vp * nk)(*[_vp(p) for p in d_sizes])
        self._chk(self.lib.mg_containment_multi_dev(ctypes.c_int(nk), c_q, c_t, ctypes.c_uint32(ci), c_h, c_s))

    def containment(self, sketch, table, ci=2):
        g = table.ngenomes
        d_h, d_s = self.empty(max(g, 1), np.uint32), self.empty(max(g, 1), np.uint32)
        try:
            self.containment_dev(sketch, table, ci, d_h.ptr, d_s.ptr)
            return d_h.download()[:g], d_s.download()[:g]
        finally:
            d_h.free()
            d_s.free()

    # ---- stage C ----
    def profile_begin_dev(self, d_recs, nrecs, has_lookahead, d_ref2tax, nref, ntax, pct_id):
        h = _vp()
        self._chk(self.lib.mg_profile_begin_dev(_vp(d_recs), ctypes.c_uint64(nrecs), ctypes.c_int(int(has_lookahead)),
                                                _vp(d_ref2tax), ctypes.c_uint32(nref), ctypes.c_uint32(ntax),
                                                ctypes.c_double(pct_id), ctypes.byref(h)))
        return ProfileShard(self, h)

    def profile_assign_resident(self, recs, ref2tax, ntax, pct_id):
        """As profile_assign, but the multimapped CSR STAYS on the device: -> (dict without the mm_* arrays, plus
        'mm_nreads' / 'mm_nentries', and 'resident' = a ResidentProfile to resolve the multimapped reads with)."""
        recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
        ref2tax = np.ascontiguousarray(ref2tax, dtype=np.uint32)
        T = max(int(ntax), 1)
        d_recs = self.array(recs if len(recs) else np.zeros(1, REC_DTYPE))
        d_r2t = self.array(ref2tax if ref2tax.size else np.zeros(1, np.uint32))
        d_acc = self.empty(3 * T + 2, np.uint64)
        shard = self.profile_begin_dev(d_recs.ptr, len(recs), False, d_r2t.ptr, len(ref2tax), ntax, pct_id)
        base = d_acc.ptr
        if len(recs):
            shard.commit(True, True, 0, base, base + 8 * T, base + 16 * T, base + 24 * T, reset=True)
            acc = d_acc.download()
            nr, ne = shard.multimapped_size()
        else:  # nothing to run: an empty stream has no boundary at all
            shard.commit(True, True, 0, base, base + 8 * T, base + 16 * T, base + 24 * T, reset=True)
            acc = d_acc.download()
            nr = ne = 0
        res = dict(count=acc[:ntax].copy(), bases=acc[T:T + ntax].copy(), first_seen=acc[2 * T:2 * T + ntax].copy(),
                   tot_rds=int(acc[3 * T]), n_ambig=int(acc[3 * T + 1]), mm_nreads=nr, mm_nentries=ne)
        res['resident'] = ResidentProfile(shard, [d_recs, d_r2t, d_acc])
        return res

    def profile_assign_dev_records(self, d_recs_ptr, n, ref2tax, ntax, pct_id, owners, resident=False):
        """Stage C over records ALREADY in HBM (the device tokeniser's batch).  owners: objects to free with the
        result (they hold the records).  resident=False: the multimapped CSR is downloaded (same dict as
        profile_assign); True: it stays on the device (same dict as profile_assign_resident)."""
        ref2tax = np.ascontiguousarray(ref2tax, dtype=np.uint32)
        T = max(int(ntax), 1)
        d_r2t = self.array(ref2tax if ref2tax.size else np.zeros(1, np.uint32))
        d_acc = self.empty(3 * T + 2, np.uint64)
        pad = None
        if not n:  # a handle needs a valid pointer even for an empty stream
            pad = self.array(np.zeros(1, REC_DTYPE))
            d_recs_ptr = pad.ptr
        shard = self.profile_begin_dev(d_recs_ptr, n, False, d_r2t.ptr, len(ref2tax), ntax, pct_id)
        base = d_acc.ptr
        shard.commit(True, True, 0, base, base + 8 * T, base + 16 * T, base + 24 * T, reset=True)
        acc = d_acc.download()
        res = dict(count=acc[:ntax].copy(), bases=acc[T:T + ntax].copy(), first_seen=acc[2 * T:2 * T + ntax].copy(),
                   tot_rds=int(acc[3 * T]), n_ambig=int(acc[3 * T + 1]))
        keep = list(owners) + [d_r2t, d_acc] + ([pad] if pad is not None else [])
        if resident:
            nr, ne = shard.multimapped_size() if n else (0, 0)
            res.update(mm_nreads=nr, mm_nentries=ne, resident=ResidentProfile(shard, keep))
            return res
        try:
            off, tax, hl, rd = shard.multimapped() if n else (np.zeros(1, np.uint64), np.zeros(0, np.uint32),
                                                             np.zeros(0, np.uint64), np.zeros(0, np.uint64))
        finally:
            shard.free()
            for b in keep:
                b.free()
        res.update(mm_offsets=off, mm_tax=tax, mm_hitlen=hl, mm_read=rd)
        return res

    def profile_assign(self, recs, ref2tax, ntax, pct_id):
        """Whole stream on one device: host arrays in, dict of host arrays out (same keys as the oracle)."""
        recs = np.ascontiguousarray(recs, dtype=REC_DTYPE)
        ref2tax = np.ascontiguousarray(ref2tax, dtype=np.uint32)
        n = len(recs)
        count = np.zeros(max(ntax, 1), dtype=np.uint64)
        bases = np.zeros(max(ntax, 1), dtype=np.uint64)
        first = np.zeros(max(ntax, 1), dtype=np.uint64)
        mm_off = np.zeros(n + 2, dtype=np.uint64)
        mm_tax = np.zeros(n + 1, dtype=np.uint32)
        mm_len = np.zeros(n + 1, dtype=np.uint64)
        mm_read = np.zeros(n + 1, dtype=np.uint64)
        tot, amb, nmm, nent = (ctypes.c_uint64(0) for _ in range(4))
        r2t = ref2tax if ref2tax.size else np.zeros(1, np.uint32)
        self._chk(self.lib.mg_profile_assign(
            _vp(recs.ctypes.data), ctypes.c_uint64(n), _np(r2t, ctypes.c_uint32), ctypes.c_uint32(len(ref2tax)),
            ctypes.c_uint32(ntax), ctypes.c_double(pct_id), _np(count, ctypes.c_uint64), _np(bases, ctypes.c_uint64),
            _np(first, ctypes.c_uint64), ctypes.byref(tot), ctypes.byref(amb), _np(mm_off, ctypes.c_uint64),
            _np(mm_tax, ctypes.c_uint32), _np(mm_len, ctypes.c_uint64), _np(mm_read, ctypes.c_uint64),
            ctypes.c_uint64(n + 1), ctypes.c_uint64(n + 1), ctypes.byref(nmm), ctypes.byref(nent)))
        m, e = nmm.value, nent.value
        return dict(count=count[:ntax], bases=bases[:ntax], first_seen=first[:ntax], tot_rds=tot.value,
                    n_ambig=amb.value, mm_offsets=mm_off[: m + 1].copy(), mm_tax=mm_tax[:e].copy(),
                    mm_hitlen=mm_len[:m].copy(), mm_read=mm_read[:m].copy())
