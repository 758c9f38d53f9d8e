#! /usr/bin/env python
"""Database pre-filter stage (drop-in for /root/reference/scripts/select_db.py).

Same command line, same function names, same files written (subset FASTA, subset db_info, and — when
not supplied with --cmash_results — temp_dir/cmash_query_results.csv in the CSV layout the reference's
cutoff logic reads, scripts/select_db.py:80-85).  What changed is how that CSV is produced:

  reference                                              here
  ---------                                              ----
  kmc -k60 -ci2 -cs3 over the reads      (:50-52)        mg_sketch_reads_dev   (k_sketch_reads)
  kmc_tools intersect with DB k-mers     (:54-56)   }    mg_containment_dev    (k_containment)
  kmc_dump + FASTA rewrite               (:58-65)   }
  StreamingQueryDNADatabase.py 30-60-10  (:73-76)   }
  cmash_db_n1000_k60.h5 / .bf / KMC dump (:44,69-70)     data/sketch_table/  (metalign_amd/formats.py)

k values, sketch size and the count threshold come from the sketch table / flags (--ks-free: the table's
own k list is used; the stock table is n=1000, k in {30,40,50,60}); containment of the LARGEST k is the last
CSV column, which is the one the cutoff applies to (:85-86).  Parity status of this arithmetic: unpinned by
the reference (KMC / CMash are not vendored), pinned to oracle/mg_oracle.c bit for bit — see DESIGN.md.
"""
import os
import sys
import tempfile

import numpy as np

from . import _hip, cli, formats


def select_parseargs(argv=None):
    return cli.parser_for('select_db').parse_args(argv)


def read_dbinfo(args):
    """{taxid: [[accessions], length_of_first_row, namelin, taxlin]} (reference :27-40)."""
    taxid2info = {}
    with open(args.dbinfo_in, 'r') as fh:
        fh.readline()
        for row in fh:
            f = row.strip().split('\t')
            entry = taxid2info.get(f[2])
            if entry is None:
                taxid2info[f[2]] = [[f[0]], f[1]] + f[3:]
            else:
                entry[0].append(f[0])
    return taxid2info


def containment_rows(names, per_k_ci):
    """CSV rows as the streaming query writes them: organisms with containment > 0 at the smallest k
    (`-c 0 --sensitive`, :75), sorted by the largest-k column, descending, ties in table order."""
    first, last = per_k_ci[0], per_k_ci[-1]
    keep = np.nonzero(first > 0)[0]
    order = keep[np.argsort(-last[keep], kind='stable')]
    return [(names[g], [float(ci[g]) for ci in per_k_ci]) for g in order]


def write_containment_csv(path, ks, rows):
    with open(path, 'w') as fh:
        fh.write(',' + ','.join('k=%d' % k for k in ks) + '\n')
        for name, vals in rows:
            fh.write(name + ',' + ','.join(repr(v) for v in vals) + '\n')


class _HostParsedReads:
    """Reads parsed on the host (multi-line FASTA) and uploaded; same surface as _hip.Reads."""

    def __init__(self, hip, bases, offsets):
        self.count = len(offsets) - 1
        self._b = hip.array(bases if bases.size else np.zeros(1, np.uint8))
        self._o = hip.array(offsets)

    def device_ptrs(self):
        return self._b.ptr, self._o.ptr

    def free(self):
        self._b.free()
        self._o.free()


def load_reads_device(hip, path, kind):
    """Reads file -> device-resident bases + offsets.  FASTQ and FASTA are parsed on the GPU from the raw
    (decompressed) text (mg_reads_parse_dev); the host parser is only the fallback for text the device rejects."""
    import gzip
    fmt = 'fastq' if kind == 'fastq' else 'fasta_ml'  # (FASTA: sequences over any number of lines)
    try:
        if path.endswith('.gz'):
            with gzip.open(path, 'rb') as fh:
                return hip.parse_reads(fh.read(), fmt)
        # plain text goes up through page-locked chunks (the file read overlaps the DMA) and is parsed where it lands
        d_text, size = hip.upload_file(path)
        try:
            return hip.parse_reads_dev(d_text.ptr, size, fmt)
        finally:
            d_text.free()
    except _hip.HipError:
        if kind == 'fastq':
            raise
        bases, offsets, _ = formats.read_sequences(path, kind)
        return _HostParsedReads(hip, bases, offsets)


# ---- one process per GPU (python -m torch.distributed.run ... -m metalign_amd.select_db ...) ----------------------------
# Every rank takes a byte range of the reads file, moved to RECORD boundaries exactly: FASTQ records are four lines, so
# a rank needs the number of newlines in front of its range (every rank counts its own range, one all-gather); FASTA
# records begin at a '>' that follows a newline.  A record belongs to the rank in whose raw range its first line
# STARTS.  The sketch tables are read by hash range (format 2), the exchange is distributed.ShardJob's.

def _scan(path, pos, size, fn, block=1 << 20):
    """fn(bytes, file offset of the block) -> file offset or None, over blocks from pos on; None at the end of file."""
    with open(path, 'rb') as fh:
        while pos < size:
            fh.seek(pos)
            buf = fh.read(block)
            if not buf:
                return None
            r = fn(buf, pos)
            if r is not None:
                return r
            pos += len(buf)
    return None


def count_newlines(path, lo, hi, block=1 << 24):
    n = 0
    with open(path, 'rb') as fh:
        fh.seek(lo)
        left = hi - lo
        while left > 0:
            buf = fh.read(min(block, left))
            if not buf:
                break
            n += int(np.count_nonzero(np.frombuffer(buf, dtype=np.uint8) == 10))
            left -= len(buf)
    return n


def fastq_record_start(path, pos, lines_before, size):
    """File offset of the first FASTQ record whose first line starts at or after byte `pos`, given the number of
    newlines in front of `pos`; `size` (end of file) when there is none."""
    if pos >= size:
        return size
    if pos == 0:
        line_start, idx = 0, 0
    else:
        with open(path, 'rb') as fh:
            fh.seek(pos - 1)
            prev = fh.read(1)
        if prev == b'\n':
            line_start, idx = pos, lines_before
        else:  # in the middle of line number `lines_before`: the next line is the first to start in here
            nl = _scan(path, pos, size, lambda buf, off: (off + buf.find(b'\n')) if b'\n' in buf else None)
            if nl is None:
                return size
            line_start, idx = nl + 1, lines_before + 1
    skip = (-idx) % 4  # lines to step over to the next record's '@' line
    while skip and line_start < size:
        nl = _scan(path, line_start, size, lambda buf, off: (off + buf.find(b'\n')) if b'\n' in buf else None)
        if nl is None:
            return size
        line_start = nl + 1
        skip -= 1
    return min(line_start, size)


def fasta_record_start(path, pos, size):
    """File offset of the first FASTA header line that starts at or after byte `pos` (0 for pos 0: what precedes the first
    header travels with the first range and is ignored by the parser, as on one GPU)."""
    if pos == 0:
        return 0
    if pos >= size:
        return size

    def find(buf, off):
        at = buf.find(b'\n>')
        if at >= 0:
            return off + at + 1
        return None
    # (blocks overlap by one byte so that a '\n' at a block's end and the '>' after it are seen together)
    p = pos - 1
    with open(path, 'rb') as fh:
        while p < size:
            fh.seek(p)
            buf = fh.read((1 << 20) + 1)
            if len(buf) < 2:
                return size
            r = find(buf, p)
            if r is not None:
                return r
            p += len(buf) - 1
    return size


def read_range_of_rank(path, kind, rank, world, all_gather_ints):
    """(start, end) byte offsets of the records of `path` that rank `rank` of `world` owns.  all_gather_ints(x) -> the
    list of every rank's x (torch.distributed in the launcher, a plain list in the tests)."""
    size = os.path.getsize(path)
    raw = [size * r // world for r in range(world + 1)]
    if kind != 'fastq':
        return fasta_record_start(path, raw[rank], size), fasta_record_start(path, raw[rank + 1], size)
    counts = all_gather_ints(count_newlines(path, raw[rank], raw[rank + 1]))
    before = [0]
    for c in counts:
        before.append(before[-1] + int(c))
    return (fastq_record_start(path, raw[rank], before[rank], size),
            fastq_record_start(path, raw[rank + 1], before[rank + 1], size))


def text_record_cuts(text, kind, world):
    """world + 1 offsets into `text` (bytes): share r = [cut[r], cut[r + 1]) holds whole records only — FASTQ records are four
    lines (a quality line may begin with '@': the line NUMBER decides), FASTA records begin at a '>' that follows a newline."""
    arr = np.frombuffer(text, dtype=np.uint8)
    size = len(arr)
    raw = [size * r // world for r in range(world + 1)]
    cuts = [0]
    lines_before = 0
    for r in range(1, world):
        lines_before += int(np.count_nonzero(arr[raw[r - 1]:raw[r]] == 10))
        pos = raw[r]
        if pos >= size:
            cuts.append(size)
            continue
        if kind != 'fastq':
            at = text.find(b'\n>', pos - 1 if pos else 0)
            cuts.append(at + 1 if at >= 0 else size)
            continue
        # the first line that STARTS at or after pos, and its number
        if pos == 0 or arr[pos - 1] == 10:
            line_start, idx = pos, lines_before
        else:
            nl = text.find(b'\n', pos)
            if nl < 0:
                cuts.append(size)
                continue
            line_start, idx = nl + 1, lines_before + 1
        for _ in range((-idx) % 4):  # on to the next record's '@' line
            nl = text.find(b'\n', line_start)
            if nl < 0:
                line_start = size
                break
            line_start = nl + 1
        cuts.append(min(line_start, size))
    cuts.append(size)
    for i in range(1, len(cuts)):  # (monotone whatever the text looks like)
        cuts[i] = max(cuts[i], cuts[i - 1])
    return cuts


def scatter_text(dist, rank, world, dev, text, kind):
    """Rank 0 holds `text` (bytes); -> this rank's record-aligned share of it.  Sizes by broadcast, the shares by point-to-point
    sends (device tensors under RCCL: the share lands in HBM over xGMI)."""
    import torch
    if world == 1:
        return text
    sizes = [None]
    if rank == 0:
        cuts = text_record_cuts(text, kind, world)
        sizes = [[cuts[r + 1] - cuts[r] for r in range(world)]]
    dist.broadcast_object_list(sizes, src=0)
    sizes = sizes[0]
    if rank == 0:
        arr = np.frombuffer(text, dtype=np.uint8)
        reqs = []
        for q in range(1, world):
            if sizes[q]:
                t = torch.from_numpy(arr[cuts[q]:cuts[q + 1]].copy()).to(dev)
                reqs.append((dist.isend(t, dst=q), t))
        for r, _ in reqs:
            r.wait()
        return text[cuts[0]:cuts[1]]
    if not sizes[rank]:
        return b''
    t = torch.empty(sizes[rank], dtype=torch.uint8, device=dev)
    dist.recv(t, src=0)
    return t.cpu().numpy().tobytes()


_dist_keep = []  # (the torch stream the library launches on must outlive the job)


def dist_context():
    """(torch.distributed, rank, world, Hip) when this process is one rank of a torch.distributed.run launch (WORLD_SIZE
    in the environment; MG_FORCE_DIST=1: also at world size 1), else None."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and os.environ.get('MG_FORCE_DIST') != '1':
        return None
    import torch
    import torch.distributed as dist
    # (MG_DIST_BACKEND=gloo — tests: several ranks on ONE GPU, which RCCL refuses; the ranks then share device
    # LOCAL_RANK modulo the device count and the collectives are staged through the host)
    local = int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if not dist.is_initialized():
        stream = torch.cuda.Stream()  # explicit: collectives are ordered with the library's kernels on it
        torch.cuda.set_stream(stream)
        _dist_keep.append(stream)
        dist.init_process_group(os.environ.get('MG_DIST_BACKEND', 'nccl'))
    hip = _hip.Hip.get(local, stream=torch.cuda.current_stream().cuda_stream)
    return dist, dist.get_rank(), dist.get_world_size(), hip


def run_sketch_steps_dist(args, ctx):
    """run_sketch_steps with one process per GPU: reads sharded by byte range, sketch tables and read sketches sharded by
    hash range (distributed.ShardJob); rank 0 writes the CSV."""
    import torch
    from .distributed import ShardJob
    dist, rank, world, hip = ctx
    table_dir = getattr(args, 'sketch_table', 'AUTO')
    if table_dir in (None, 'AUTO'):
        table_dir = formats.default_table_dir(args.data)
    table = formats.SketchTable(table_dir)
    hip.set_hash_mode(table.hash_mode)

    dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'

    def gather(x):
        t = torch.tensor([int(x)], dtype=torch.int64, device=dev)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [int(o.item()) for o in out]

    if formats.is_gzip(args.reads):
        # rank 0 inflates the file ON ITS GPU (mg_inflate.hip: compressed bytes up, text down; round 4 did it with every core of
        # the host, mg_pgzip.hip — 0.40 s against 0.07 + 0.06 s for a 10M-read file) and SCATTERS record-aligned shares of the
        # text: every rank parses and sketches its own (rounds 2-3: rank 0 inflated on one core and sketched everything, the
        # others brought empty shards)
        whole = None
        if rank == 0:
            try:
                with open(args.reads, 'rb') as fh:
                    whole = hip.inflate(fh.read())
            except _hip.HipError as e:
                if e.code != _hip.ERR_NOMEM:
                    raise
                # (the device has no room for the compressed file and its text at once: the host inflater, one piece at a time)
                whole = _hip.gunzip_file(args.reads)
        text = scatter_text(dist, rank, world, dev, whole, args.input_type)
        del whole
    else:
        start, end = read_range_of_rank(args.reads, args.input_type, rank, world, gather)
        with open(args.reads, 'rb') as fh:
            fh.seek(start)
            text = fh.read(end - start)
    reads = hip.parse_reads(text, 'fastq' if args.input_type == 'fastq' else 'fasta_ml')
    rb, ro = reads.download()
    reads.free()
    if os.environ.get('MG_DIST_REPORT') == '1':  # (tests: what every rank's shard held)
        with open(os.path.join(args.temp_dir, 'shard_rank%d.txt' % rank), 'w') as fh:
            fh.write('%d reads, %d bases\n' % (len(ro) - 1, len(rb)))
    # (a reference-pipeline table, formats.py version 3: the job sketches the largest k only and shares the table out by hash
    # range for the pairs and by prefix range for the smaller k's count lists)
    job = ShardJob(hip, dist, rank, world, k=list(table.ks), ci=int(getattr(args, 'min_count', 2)),
                   s=int(getattr(args, 'sketch_size', 0)), always_exchange=True, definition=table.definition)
    # (a format-2 table is hash-major on disk: the rank maps only its hash range; a format-1 table is inverted on the host
    # first, by every rank)
    job.load(rb, ro, np.zeros(0, dtype=_hip.REC_DTYPE), np.zeros(1, dtype=np.uint32), table, ntax=1)
    got = job.step()
    out = args.temp_dir + 'cmash_query_results.csv'
    if rank == 0:
        per_k = []
        for hits, sizes in zip(got['hits_k'], got['sizes_k']):
            with np.errstate(divide='ignore', invalid='ignore'):
                per_k.append(np.where(sizes > 0, hits.astype(np.float64) / sizes.astype(np.float64), 0.0))
        write_containment_csv(out, table.ks, containment_rows(table.names, per_k))
    dist.barrier()
    return out


def _record_cut(data, kind):
    """Index just past the last COMPLETE record of a piece of FASTQ / FASTA text (0: none yet)."""
    arr = np.frombuffer(data, dtype=np.uint8)
    if kind == 'fastq':  # four lines per record (what the device parser takes)
        nl = np.flatnonzero(arr == 10)
        whole = (len(nl) // 4) * 4
        return int(nl[whole - 1]) + 1 if whole else 0
    at = data.rfind(b'\n>')  # FASTA: a record ends where the next header begins
    return at + 1 if at >= 0 else 0


def iter_read_batches(hip, path, kind, batch_bytes):
    """Device-resident batches of the reads file: the whole file when it fits `batch_bytes` (plain text goes up through
    page-locked chunks and is parsed where it lands), otherwise record-aligned pieces of about that size — a sample
    larger than the device's free memory is sketched piece by piece and the sketches are merged (run_sketch_steps);
    the reference's kmc spills to disk instead (scripts/select_db.py:50-52)."""
    import gzip
    gz = path.endswith('.gz')
    if not gz and os.path.getsize(path) <= batch_bytes:
        yield load_reads_device(hip, path, kind)
        return
    fmt = 'fastq' if kind == 'fastq' else 'fasta_ml'
    carry = b''
    piece = min(batch_bytes, 2 << 30)  # (a piece passes through host memory: bounded whatever the device could take)
    with (gzip.open(path, 'rb') if gz else open(path, 'rb')) as fh:
        while True:
            buf = fh.read(piece)
            data = carry + buf
            if not buf:  # end of file: what is left is the last record (with or without a final newline)
                if data.strip():
                    yield hip.parse_reads(data, fmt)
                return
            cut = _record_cut(data, kind)
            if cut:
                yield hip.parse_reads(data[:cut], fmt)
            carry = data[cut:]


def expected_bases(path, kind):
    """Roughly the bases in a reads file, from its size: sizes the streamed sketch's counting tables (an estimate that
    proves too small is caught — a table overflow — and the file streamed again).  FASTQ: half the bytes are bases;
    gzip: DNA text deflates about 4 x."""
    size = os.path.getsize(path)
    if path.endswith('.gz'):
        guess = size * 4
        try:  # a gzip member ends with the length of its text (mod 2^32): the whole text for a file of one member below 4 GB
            with open(path, 'rb') as fh:
                fh.seek(-4, os.SEEK_END)
                isize = int.from_bytes(fh.read(4), 'little')
            if size < isize < 64 * size:
                guess = max(guess if isize < size * 2 else 0, isize)
        except OSError:
            pass
        size = guess
    return int(size * (0.5 if kind == 'fastq' else 1.0)) + 1


def stream_reads_file(hip, path, kind, ks, hmaxs, s, filts, offset=0, length=0):
    """The reads file -> read sketches of every k, streamed: reader threads fill page-locked chunks (plain files by
    positional reads in parallel, `.gz` inflated by zlib inside the library — BGZF blocks in parallel), chunk i + 1 goes up
    while chunk i is parsed ON THE DEVICE and hashed into ONE set of counting tables (mg_sketch_stream_add_file): no
    per-chunk sketch, no merge, and the text never exists as a host array.  What kmc does with the reads file,
    scripts/select_db.py:45-52 (`.gz` expected: :146-148).
    None when the file does not suit the pipeline (a record larger than a chunk's headroom; FASTA text the device parser
    rejects): the caller takes the piece-wise path, whose host parser decides."""
    fmt = 'fastq' if kind == 'fastq' else 'fasta_ml'
    expect = expected_bases(path, kind) if not length else int(length * (0.5 if kind == 'fastq' else 1.0)) + 1
    chunk = int(os.environ.get('MG_STREAM_CHUNK_BYTES', 0))
    for attempt in range(2):
        stream = hip.sketch_stream(ks, hmaxs, s, filts, expect)
        sks = []
        try:
            try:
                stream.add_file(path, fmt, offset=offset, length=length, chunk_bytes=chunk)
            except _hip.HipError as e:
                # a record longer than a piece's headroom (capacity), or FASTA text the device parser refuses: not for
                # this pipeline; a malformed FASTQ record is the caller's error, as on the whole-file path
                if e.code == _hip.ERR_CAPACITY or (e.code == _hip.ERR_ARG and kind != 'fastq'):
                    return None
                raise
            sks = stream.finish()
            try:
                for sk in sks:
                    sk.resolve()
                return sks
            except _hip.HipError as e:
                for sk in sks:
                    sk.free()
                if e.code != _hip.ERR_CAPACITY:
                    raise
                # a counting table overflowed: the library has reset its hint to the worst case; size the second pass
                # for the bases the first one counted
                expect = max(expect, int(stream.nbases * 1.25))
        finally:
            stream.free()
    return None


def _merge_two(hip, a, b, k, hmax, s):
    """Union of two read sketches of the same k with saturating count sums (mg_sketch_merge_dev: the multi-GPU merge,
    here for the pieces of one sample).  Frees both inputs."""
    (ha, ca), (hb, cb) = a.download(), b.download()
    trunc = [(sk.truncated, sk.last_hash) for sk in (a, b)]
    any_trunc = any(t for t, _ in trunc)
    bound = min([lh for t, lh in trunc if t] or [_hip.U64_MAX])
    a.free()
    b.free()
    d_h, d_c = hip.array(np.concatenate([ha, hb])), hip.array(np.concatenate([ca, cb]))
    try:
        return hip.sketch_merge_dev(d_h.ptr, d_c.ptr, len(ha) + len(hb), k, 0, hmax, s, any_trunc, bound)
    finally:
        d_h.free()
        d_c.free()


run_timings = {}  # seconds of the last select_main's parts (tools/bench_cli.py reads them; nothing else does)


def run_sketch_steps(args):
    """Stages A+B on the MI355X: reads -> per-k read sketch -> containment of every genome sketch ->
    temp_dir/cmash_query_results.csv.  Replaces run_kmc_steps (:43-65) and the CMash call (:69-76)."""
    import time
    t_start = time.perf_counter()
    hip = _hip.Hip.get()
    table_dir = getattr(args, 'sketch_table', 'AUTO')
    if table_dir in (None, 'AUTO'):
        table_dir = formats.default_table_dir(args.data)
    table = formats.SketchTable(table_dir)
    previous_mode = hip.hash_mode
    hip.set_hash_mode(table.hash_mode)  # the reads are hashed by the definition the table was sketched with
    try:
        return _run_sketch_steps(args, hip, table, t_start)
    finally:
        hip.set_hash_mode(previous_mode)


def stream_ok(ks):
    """Whether mg_sketch_stream_begin takes this k set (1..4 of them, ascending); any other set goes through the piece-wise
    path, whose per-k launches take any number of k in any order."""
    return 1 <= len(ks) <= 4 and all(a < b for a, b in zip(ks, ks[1:]))


def _run_sketch_steps(args, hip, table, t_start):
    import time
    min_count = int(getattr(args, 'min_count', 2))
    s = int(getattr(args, 'sketch_size', 0))
    # A reference-pipeline table (formats.py, version 3): the reads are sketched at the LARGEST k only — what the reference's
    # kmc call counts (:50-52) — and stage B derives the smaller k's columns from the matched k_max-mers (:54-59, :73-76).
    # Any other table: every k of the table has its own hash-major pairs, which go up as they lie on disk (no sort) ...
    reftable = None
    if table.refpipe:
        if s:
            sys.exit('Error: a reference-pipeline sketch table counts every k_max-mer of the reads; --sketch_size does not apply.')
        arrays = table.refpipe_arrays()
        if kmer_match_applies(args, table, arrays):
            return _run_count_steps(args, hip, table, arrays, t_start)
        # (the table goes up BESIDE the reads: the library's uploader threads copy the memory maps through page-locked slots of their
        # own while the stream below runs — 450 MB at 10k genomes, 20 ms of select_main that used to come first; stage B waits for it)
        reftable = hip.refdb_upload(arrays['ks'], arrays['ngenomes'], arrays['pair_hash'], arrays['pair_gen'], arrays['gsize'],
                                    arrays['max_hash'], arrays['small'], wait=False)
        sketch_ks, dev_tables = [table.ks[-1]], []
    else:
        sketch_ks = list(table.ks)
        dev_tables = [hip.upload_table_sorted(**table.pairs(k)) for k in sketch_ks]
    # ... the stored pre-filter with them (the role of the reference's bloom pre-filter, -f ...bf, :70,75) ...
    filts = []
    for k in sketch_ks:
        bits = table.filter_bits(k)
        filts.append(hip.filter_from_bits(bits) if bits is not None
                     else hip.filter_build(np.asarray(table.pairs(k)['pair_hash'])))
    # ... then ONE pass over the reads for all k (the reference's query is multi-k too: 30-60-10, :75), and stage B per k
    # (a reads file larger than a quarter of the free device memory — or MG_READ_BATCH_BYTES — goes through in
    # record-aligned pieces whose sketches are merged: saturating counters add up to the same clamped counts)
    hmaxs = [t.max_hash for t in dev_tables] if reftable is None else [reftable.max_hash]
    run_timings['table_load_s'] = time.perf_counter() - t_start
    t_start = time.perf_counter()
    sks = None
    if os.environ.get('MG_NO_STREAM') != '1' and stream_ok(sketch_ks):
        sks = stream_reads_file(hip, args.reads, args.input_type, sketch_ks, hmaxs, s, filts)
    free, _, pooled = hip.mem_info()
    batch_bytes = int(os.environ.get('MG_READ_BATCH_BYTES', 0)) or max((free + pooled) // 4, 1 << 26)
    for reads in (iter_read_batches(hip, args.reads, args.input_type, batch_bytes) if sks is None else ()):
        d_b_ptr, d_o_ptr = reads.device_ptrs()
        if sks is None and reads.count > int(os.environ.get('MG_PRIME_READS', 4_000_000)):
            # the library sizes a k's counting table from the distinct-to-candidate ratio of its previous call and has
            # none yet (worst case: tens of GB to clear and sort against a dense table): the first million reads tell it
            for sk in hip.sketch_reads_multi_dev_async(d_b_ptr, d_o_ptr, min(1_000_000, max(reads.count // 4, 1)), sketch_ks,
                                                       hmaxs, s, filts):
                sk.resolve()
                sk.free()
        part = hip.sketch_reads_multi_dev_async(d_b_ptr, d_o_ptr, reads.count, sketch_ks, hmaxs, s, filts)
        for sk in part:
            sk.resolve()  # (a sketch whose counting table overflowed is redone from the reads: before they go)
        reads.free()
        sks = part if sks is None else [_merge_two(hip, a, b, k, hm, s) for a, b, k, hm in zip(sks, part, sketch_ks, hmaxs)]
    if sks is None:  # an empty reads file (on the piece-wise path)
        empty = _HostParsedReads(hip, np.zeros(0, np.uint8), np.zeros(1, np.uint64))
        sks = hip.sketch_reads_multi_dev_async(*empty.device_ptrs(), 0, sketch_ks, hmaxs, s, filts)
        for sk in sks:
            sk.resolve()
        empty.free()
    run_timings['stream_s'] = time.perf_counter() - t_start
    t_start = time.perf_counter()
    per_k = []
    with np.errstate(divide='ignore', invalid='ignore'):
        if reftable is not None:
            hits_k, sizes_k = hip.refpipe_containment(sks[0], reftable, min_count)
            for hits, sizes in zip(hits_k, sizes_k):
                per_k.append(np.where(sizes > 0, hits.astype(np.float64) / sizes.astype(np.float64), 0.0))
        else:
            for sk, dev_table in zip(sks, dev_tables):
                hits, sizes = hip.containment(sk, dev_table, min_count)
                per_k.append(np.where(sizes > 0, hits.astype(np.float64) / sizes.astype(np.float64), 0.0))
    for h in sks + filts + dev_tables + ([reftable] if reftable is not None else []):
        h.free()
    out = args.temp_dir + 'cmash_query_results.csv'
    write_containment_csv(out, table.ks, containment_rows(table.names, per_k))
    run_timings['containment_s'] = time.perf_counter() - t_start
    return out


def kmer_match_applies(args, table, arrays):
    """Stage A BY K-MER IDENTITY (mg_kcount.hip; the default): the reads' k_max-mers are counted among the table's as `kmc` +
    `kmc_tools intersect` do it (:50-59) — as k-mers, nothing on the read side hashed.  Needs the table's k-mers (format 3 stores
    them) and 15 <= k_max <= 64; `--kmer_match hash` keeps the read sketch of rounds 4-5."""
    from .distributed import kmer_match_by_default
    want = str(getattr(args, 'kmer_match', 'identity'))
    if want == 'hash':
        return False
    ok = arrays.get('kmer_hi') is not None and 15 <= table.ks[-1] <= 64
    if not ok and want == 'identity_only':
        sys.exit('Error: --kmer_match identity_only needs a reference-pipeline table that stores its k-mers, with the largest k in [15, 64].')
    # (identity: where it is the faster of the two — from k_max = 25 on; identity_only: wherever it can run)
    return ok and (want == 'identity_only' or kmer_match_by_default(table.ks[-1]))


def _run_count_steps(args, hip, table, arrays, t_start):
    """run_sketch_steps for a reference-pipeline table, k-mers met by identity: table + its k-mer index up, the reads file
    streamed through the device parser into ONE set of counters (no sketch, no merge of pieces), stage B from the counters."""
    import time
    min_count = int(getattr(args, 'min_count', 2))
    reftable = hip.refdb_upload(arrays['ks'], arrays['ngenomes'], arrays['pair_hash'], arrays['pair_gen'], arrays['gsize'],
                                arrays['max_hash'], arrays['small'], wait=False)
    reftable.index_kmers(arrays['kmer_hi'], arrays['kmer_lo'])
    counts = reftable.kmer_counts()
    run_timings['table_load_s'] = time.perf_counter() - t_start
    t_start = time.perf_counter()
    kind = args.input_type
    done = False
    if os.environ.get('MG_NO_STREAM') != '1':
        stream = hip.count_stream(counts)
        try:
            stream.add_file(args.reads, 'fastq' if kind == 'fastq' else 'fasta_ml', chunk_bytes=int(os.environ.get('MG_STREAM_CHUNK_BYTES', 0)))
            done = True
        except _hip.HipError as e:
            # a record longer than a piece's headroom (capacity), or FASTA text the device parser refuses: the piece-wise path's
            # host parser decides; a malformed FASTQ record is the caller's error, as on the whole-file path
            if not (e.code == _hip.ERR_CAPACITY or (e.code == _hip.ERR_ARG and kind != 'fastq')):
                raise
            counts.reset()
        finally:
            stream.free()
    if not done:
        free, _, pooled = hip.mem_info()
        batch_bytes = int(os.environ.get('MG_READ_BATCH_BYTES', 0)) or max((free + pooled) // 4, 1 << 26)
        for reads in iter_read_batches(hip, args.reads, kind, batch_bytes):
            counts.add_reads(reads)
            hip.sync()  # (the batch's buffers go back to the pool below)
            reads.free()
    run_timings['stream_s'] = time.perf_counter() - t_start
    t_start = time.perf_counter()
    per_k = []
    with np.errstate(divide='ignore', invalid='ignore'):
        hits_k, sizes_k = hip.refpipe_containment_counts(counts, reftable, min_count)
        for hits, sizes in zip(hits_k, sizes_k):
            per_k.append(np.where(sizes > 0, hits.astype(np.float64) / sizes.astype(np.float64), 0.0))
    counts.free()
    reftable.free()
    out = args.temp_dir + 'cmash_query_results.csv'
    write_containment_csv(out, table.ks, containment_rows(table.names, per_k))
    run_timings['containment_s'] = time.perf_counter() - t_start
    return out


run_kmc_steps = run_sketch_steps  # the reference's name for this step


def run_cmash_and_cutoff(args, taxid2info):
    """CSV -> organisms to align against (reference :68-96; cutoff on the LAST column, first strain per
    species unless --strain_level, empty species never deduplicated)."""
    cmash_out = args.temp_dir + 'cmash_query_results.csv' if args.cmash_results == 'NONE' else args.cmash_results
    chosen, seen_species = [], set()
    with open(cmash_out, 'r') as fh:
        fh.readline()
        for row in fh:
            f = row.strip().split(',')
            organism, ci = f[0], float(f[-1])
            if not ci >= args.cutoff:
                continue
            if not args.strain_level:
                taxid = organism.split('taxid_')[1].split('_genomic.fna')[0].replace('_', '.')
                species = taxid2info[taxid][3].split('|')[-2]
                if species in seen_species and species != '':
                    continue
                seen_species.add(species)
            chosen.append(organism)
    return chosen


def _zcat_into(out_path, paths, threads=0):
    """`zcat` of every path into out_path (created / truncated), in order: the library's host threads inflate the files and
    write them at their offsets (mg_zcat_files).  The per-file work in Python threads — open, read, zlib, write — held the
    interpreter lock for most of its 0.05 s on 500 genomes, a quarter of select_main at 10M reads."""
    from . import _hip
    _hip.zcat_files(list(paths), out_path, threads)


def make_db_and_dbinfo(args, organisms_to_include, taxid2info):
    """Concatenate the selected genomes and write the subset db_info (reference :99-117)."""
    # The reference truncates args.db and starts one `zcat` per genome that appends its output (:101-105; exit codes ignored).
    # The bytes written here are the same — every selected file inflated (all members of it) in the reference's order — by
    # zlib in host threads of the library: a file zcat would refuse (not gzip) contributes nothing and a line on stderr.
    _zcat_into(args.db, [args.db_dir + o for o in organisms_to_include])
    with open(args.dbinfo_out, 'w') as out:
        out.write('Accesion\tLength\tTaxID\tLineage\tTaxID_Lineage\n')  # sic: the reference's header
        out.write('Unmapped\t0\tUnmapped\t|||||||Unmapped\t|||||||Unmapped\n')
        for organism in organisms_to_include:
            taxid = organism.split('taxid_')[1].split('_genomic.fna')[0].replace('_', '.')
            accs, length, namelin, taxlin = taxid2info[taxid][:4]
            for acc in accs:
                out.write('\t'.join([acc, length, taxid, namelin, taxlin]) + '\n')


def select_main(args=None):
    if args is None:
        args = select_parseargs()
    elif args.cutoff < 0.0 or args.cutoff > 1.0:
        print('Error: args.cutoff must be between 0 and 1, inclusive.')
        sys.exit()
    args.data = cli.with_slash(args.data)
    if args.db_dir == 'AUTO':
        args.db_dir = args.data + 'organism_files/'
    args.db_dir = cli.with_slash(args.db_dir)
    if args.temp_dir == 'AUTO/':
        args.temp_dir = tempfile.mkdtemp(prefix=args.data)
    args.temp_dir = cli.with_slash(args.temp_dir)
    os.makedirs(args.temp_dir, exist_ok=True)
    for attr, default in (('dbinfo_in', args.data + 'db_info.txt'), ('dbinfo_out', args.temp_dir + 'subset_db_info.txt'),
                          ('db', args.temp_dir + 'cmashed_db.fna')):
        if getattr(args, attr) == 'AUTO':
            setattr(args, attr, default)
    if args.input_type == 'AUTO':
        args.input_type = cli.sniff_reads_type(args.reads)

    # db_info is parsed by a second thread while the reads stream through the device (the library calls release the
    # interpreter lock): 0.011 s of a 0.16 s select_main at 10M reads.  The file is opened here, so that a missing one still
    # fails before anything is sketched, as in the reference (:143).
    open(args.dbinfo_in, 'r').close()
    from concurrent.futures import ThreadPoolExecutor
    dbinfo_pool = ThreadPoolExecutor(1)
    dbinfo_job = dbinfo_pool.submit(read_dbinfo, args)
    dbinfo_pool.shutdown(wait=False)
    ctx = dist_context() if args.cmash_results == 'NONE' else None
    if ctx is not None:  # one process per GPU: all ranks sketch, rank 0 goes on alone
        run_sketch_steps_dist(args, ctx)
        if ctx[1] != 0:
            return
    elif int(os.environ.get('WORLD_SIZE', '1')) > 1 and int(os.environ.get('RANK', '0')) != 0:
        # a multi-GPU launch with --cmash_results given: nothing to sketch, and the host-only tail below writes
        # args.db / args.dbinfo_out — ONE writer, rank 0 (W ranks appending zcat output to the same file interleave)
        return
    elif args.cmash_results == 'NONE':
        run_sketch_steps(args)
    import time
    t_tail = time.perf_counter()
    taxid2info = dbinfo_job.result()
    organisms = run_cmash_and_cutoff(args, taxid2info)
    make_db_and_dbinfo(args, organisms, taxid2info)
    run_timings['host_tail_s'] = time.perf_counter() - t_tail
    # the reference removes its KMC intermediates here (:161-167); this path creates none.
    # cmash_query_results.csv stays in temp_dir, as it does in the reference.


if __name__ == '__main__':
    select_main(select_parseargs())
