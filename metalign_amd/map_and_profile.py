#! /usr/bin/env python
"""Map + profile stage (drop-in for /root/reference/scripts/map_and_profile.py).

Same command line, same function names and return shapes as the reference
module, so `metalign.py` and third-party callers can swap it in:

    get_acc2info(args)                        scripts/map_and_profile.py:64-81
    map_and_process(args, instream, a2i, t2i) :193-264   <- runs on the MI355X
    preprocess_multimapped / resolve_multi_prop / tree_results_cami /
    compute_abundances / gather_results / write_results / map_main

What changed: the per-line / per-read Python loop of map_and_process is
replaced by (1) a tokeniser that turns each retained SAM line into a 16-byte
record and (2) the HIP kernels behind `mg_profile_*` (metalign_amd/csrc/
mg_profile.hip), which reproduce the reference's read classification —
carried "drop the next first line" state, phantom first boundary and
unflushed last read included.  Everything after the loop is O(#taxa) host
arithmetic kept in double precision in the reference's operation order so the
CAMI file is byte-identical.

There is no CPU fallback: without libmetalign_hip.so / a GPU, map_and_process
raises (metalign_amd._hip.HipUnavailable).
"""
import os
import subprocess
import sys
import time

import numpy as np

from . import _hip, cli

start = time.time()
RANKS = ['superkingdom', 'phylum', 'class', 'order', 'family', 'genus', 'species', 'strain']
_RANK_POS = {r: i for i, r in enumerate(RANKS)}


def echo(msg, verbose):
    if not verbose:
        return
    elapsed = int(time.time() - start)
    print('[%02d:%02d:%02d] %s' % (elapsed // 3600, (elapsed // 60) % 60, elapsed % 60, msg))


def profile_parseargs(argv=None):
    return cli.parser_for('map_and_profile').parse_args(argv)


def get_taxid_rank(taxlin):
    """Rank named by the last non-empty field of a 7-pipe lineage (reference :49-57)."""
    fields = taxlin.split('|')
    trailing = 0
    for i in range(1, len(taxlin) + 1):  # the reference bounds this walk by the STRING length
        if fields[-i] != '':
            break
        trailing += 1
    return RANKS[-(trailing + 1)]


def get_acc2info(args):
    """db_info -> (acc2info, taxid2info) exactly as the reference builds them (:64-81)."""
    echo('Reading dbinfo file...', args.verbose)
    acc2info, taxid2info = {}, {}
    rank_of = {}  # (a genome's accessions share their lineage: the rank of a lineage string is worked out once)
    with open(args.dbinfo, 'r') as fh:
        fh.readline()
        for row in fh:
            acc, acclen, taxid, namelin, taxlin = row.strip().split('\t')
            rank = rank_of.get(taxlin)
            if rank is None:
                rank = rank_of[taxlin] = get_taxid_rank(taxlin)
            if rank == 'strain' and acc != 'Unmapped':
                taxid, taxlin = taxid + '.1', taxlin + '.1'
            acclen = int(acclen)
            acc2info[acc] = [acclen, taxid, namelin, taxlin]
            if taxid in taxid2info:
                taxid2info[taxid][0] += acclen
            else:
                taxid2info[taxid] = [acclen, rank, namelin, taxlin]
    return acc2info, taxid2info


# --------------------------------------------------------------------------------------
# SAM text -> 16-byte records (include/metalign_hip.h: mg_aln_rec)
# --------------------------------------------------------------------------------------
class _Tokeniser:
    """Line filter of :201-217 + CIGAR walk of :86-100, producing one record per retained line.

    Errors are raised eagerly, with the exception types the reference raises lazily for the same
    input (KeyError for an unknown RNAME :217, IndexError for < 12 fields :97, ValueError for a
    CIGAR the reference cannot parse :90-93).
    """

    def __init__(self, acc_index):
        self.acc_index = acc_index
        self.prev = ''
        self.rows = []

    def feed(self, line):
        if line.startswith('@'):
            return
        f = line.strip().split()
        if len(f) < 6:
            return
        flag = int(f[1])
        cigar = f[5]
        if (flag & 4) or cigar == '*':
            return
        ref = self.acc_index[f[2]]
        matched = total = num = 0
        for ch in cigar:
            if ch.isalpha():
                if ch == 'M':
                    matched += num
                total += num
                num = 0
            else:
                num = num * 10 + int(ch)  # '=' lands here and raises ValueError, as in the reference
        int(f[11][5:])
        if total == 0:
            raise ZeroDivisionError('float division by zero')
        seqlen = 0 if f[9] == '*' else len(f[9])
        if seqlen > _hip.MAX_SEQLEN or matched > 0xFFFFFFFF or total > 0xFFFFFFFF:
            raise OverflowError('alignment longer than the record format allows')
        new = f[0] != self.prev
        self.prev = f[0]
        self.rows.append((ref | (_hip.NEW_BIT if new else 0), matched, total,
                          (flag & 0xFFF) | (seqlen << _hip.LEN_SHIFT)))

    def records(self):
        if not self.rows:
            return np.zeros(0, dtype=_hip.REC_DTYPE)
        return np.array(self.rows, dtype=_hip.REC_DTYPE)


def tokenise_sam(lines, acc_index, decode=False):
    tk = _Tokeniser(acc_index)
    for line in lines:
        if isinstance(line, (bytes, bytearray)):  # the aligner's pipe, or a SAM file opened in binary mode
            line = line.decode('utf-8')
            if decode and not line:
                break
        tk.feed(line)
    return tk.records()


def tokenise_paf(lines, acc_index, decode=False):
    """PAF replay adaptor: minimap2 PAF lines -> the same 16-byte records.

    The reference has no PAF parser (it reads SAM columns, :87,97,142-144,211,217), so this is an adaptor, not a
    parity path.  Mapping: RNAME <- target name (col 6); FLAG <- 16 if strand '-', + 256 if `tp:A:S`
    (secondary), + 2048 if `tp:A:I`/`tp:A:i` is absent and the line repeats a primary (not emitted by minimap2);
    CIGAR totals <- `cg:Z:` when present (matched = sum of M, total = all ops + the query bases outside
    [qstart, qend), i.e. what SAM writes as clipping), else matched = col 10 (matching bases) and
    total = col 2 (query length) — without `-c` PAF carries no CIGAR, so filter_line (:86-100) can only be
    approximated; len(SEQ) <- query length for primaries, 0 for secondaries (SAM writes '*' there).
    Pair flags do not exist in PAF: every read is treated as single-end."""
    return _tokenise_paf_from(lines, acc_index, '', decode)[0]


def _tokenise_paf_from(lines, acc_index, prev, decode=False):
    """tokenise_paf with the previous retained QNAME handed in and out: (records, last retained QNAME)."""
    rows = []
    for line in lines:
        if isinstance(line, (bytes, bytearray)):  # a file opened in binary mode (map_main does), or the aligner's pipe
            line = line.decode('utf-8')
            if decode and not line:
                break
        f = line.rstrip('\r\n').split('\t')
        if len(f) < 12:
            continue
        qlen, qs, qe = int(f[1]), int(f[2]), int(f[3])
        tags = {t[:4]: t[5:] for t in f[12:] if len(t) > 5}
        flag = 16 if f[4] == '-' else 0
        secondary = tags.get('tp:A') == 'S'
        if secondary:
            flag |= 256
        if 'cg:Z' in tags:
            matched = total = num = 0
            for ch in tags['cg:Z']:
                if ch.isdigit():
                    num = num * 10 + int(ch)
                else:
                    if ch == 'M':
                        matched += num
                    total += num
                    num = 0
            total += qlen - (qe - qs)
        else:
            matched, total = int(f[9]), qlen
        if total == 0:
            raise ZeroDivisionError('float division by zero')
        seqlen = 0 if secondary else qlen
        new = f[0] != prev
        prev = f[0]
        rows.append((acc_index[f[5]] | (_hip.NEW_BIT if new else 0), matched, total,
                     flag | (seqlen << _hip.LEN_SHIFT)))
    return (np.array(rows, dtype=_hip.REC_DTYPE) if rows else np.zeros(0, dtype=_hip.REC_DTYPE)), prev


_CHUNK_BYTES = 256 << 20


def _line_chunks(instream, decode):
    """The SAM stream as byte chunks that end on line boundaries.  A (binary) file object is read in bulk and
    sliced without copying; an iterator of lines (the aligner's pipe) is batched."""
    if hasattr(instream, 'read') and not decode:
        carry = b''
        while True:
            blk = instream.read(1 << 30)
            if not blk:
                break
            if isinstance(blk, str):  # a text-mode handle: the reference opens SAM files that way
                blk = blk.encode('utf-8')
            if carry:
                blk = carry + blk
            view = memoryview(blk)
            pos, n = 0, len(blk)
            while n - pos > _CHUNK_BYTES:
                cut = blk.rfind(b'\n', pos, pos + _CHUNK_BYTES) + 1
                if cut <= pos:  # a single line longer than a chunk
                    cut = blk.find(b'\n', pos + _CHUNK_BYTES) + 1
                    if cut <= 0:
                        break
                yield view[pos:cut]
                pos = cut
            last = blk.rfind(b'\n', pos) + 1
            if last > pos:
                yield view[pos:last]
                pos = last
            carry = bytes(view[pos:])
        if carry:
            yield carry
        return
    buf, size = [], 0
    for line in instream:
        if decode:
            if not line:
                break
        else:
            line = line.encode('utf-8') if isinstance(line, str) else line
        buf.append(line)
        size += len(line)
        if size >= _CHUNK_BYTES:
            yield b''.join(buf)
            buf, size = [], 0
    if buf:
        yield b''.join(buf)


def tokenise_sam_device(instream, acc_index, decode=False, paf=False):
    """SAM (paf=True: PAF) stream -> records on the MI355X (mg_sam_tokenize / mg_paf_tokenize).  A line the reference
    cannot parse is re-run through the host tokeniser so that the very same exception (type and message) surfaces."""
    hip = _hip.Hip.get()
    names = [None] * len(acc_index)
    for a, i in acc_index.items():
        names[i] = a
    index = hip.acc_index(names)
    parts, prev = [], ''
    try:
        for chunk in _line_chunks(instream, decode):
            try:
                recs, prev = hip.sam_tokenize(chunk, index, prev, paf=paf)
            except _hip.SamParseError as e:
                lines = bytes(chunk).split(b'\n')
                bad = lines[e.line].decode('utf-8', 'replace')
                if paf:
                    tokenise_paf([bad], acc_index)  # raises KeyError / ValueError / ZeroDivisionError
                else:
                    _Tokeniser(acc_index).feed(bad)  # raises KeyError / IndexError / ValueError / ZeroDivisionError
                # The host statement takes the line: the device parser is the stricter of the two (Python's int() accepts
                # surrounding blanks, '_' between digits and values of any size; the kernel does not).  The host
                # tokeniser is the definition — this chunk goes through it, carrying the previous QNAME in and out.
                if lines and not lines[-1]:
                    lines.pop()  # (the chunk ends in a newline)
                if paf:
                    recs, prev = _tokenise_paf_from(lines, acc_index, prev)
                else:
                    tk = _Tokeniser(acc_index)
                    tk.prev = prev
                    for ln in lines:
                        tk.feed(ln.decode('utf-8'))
                    recs, prev = tk.records(), tk.prev
            parts.append(recs)
    finally:
        index.free()
    if not parts:
        return np.zeros(0, dtype=_hip.REC_DTYPE)
    return parts[0] if len(parts) == 1 else np.concatenate(parts)


_device_tokenise = tokenise_sam_device


def tokenise_paf_device(instream, acc_index, decode=False):
    """PAF replay on the device (mg_paf_tokenize): the same records as tokenise_paf."""
    return tokenise_sam_device(instream, acc_index, decode=decode, paf=True)


_device_tokenise_paf = tokenise_paf_device


def dense_tables(acc2info, taxid2info):
    """Dense ids for the device: accession row -> taxon row."""
    taxids = list(taxid2info)
    tax_index = {t: i for i, t in enumerate(taxids)}
    acc_index = {a: i for i, a in enumerate(acc2info)}
    ref2tax = np.fromiter((tax_index[v[1]] for v in acc2info.values()), dtype=np.uint32, count=len(acc2info))
    return acc_index, taxids, ref2tax


def _device_assign(recs, ref2tax, ntax, pct_id):
    return _hip.Hip.get().profile_assign(recs, ref2tax, ntax, pct_id)


_DEVICE_ASSIGN = _device_assign


def _device_assign_resident(recs, ref2tax, ntax, pct_id):
    """--device_multimap: the multimapped CSR stays in HBM (res['resident']) for resolve_multi_prop_device."""
    return _hip.Hip.get().profile_assign_resident(recs, ref2tax, ntax, pct_id)


def multimapped_lists(res, taxids):
    """Multimapped CSR -> the reference's list of [taxid, ..., hitlen] lists (:245-248)."""
    off, mtax, mlen = res['mm_offsets'], res['mm_tax'], res['mm_hitlen']
    names = np.asarray(taxids, dtype=object)[mtax] if len(mtax) else []
    out = []
    for i in range(len(mlen)):
        row = list(names[int(off[i]):int(off[i + 1])])
        row.append(int(mlen[i]))
        out.append(row)
    return out


def assemble_taxids2abs(args, res, taxids, taxid2info, want_lists=True):
    """Kernel outputs -> the (taxids2abs, multimapped, low_mem_mmap) triple of the reference (:193-264).
    want_lists=False leaves `multimapped` as the CSR dict for the vectorised tail (compute_abundances)."""
    taxids2abs = {'Unmapped': [0.0, 0.0] + taxid2info['Unmapped']}
    tot_rds, n_ambig = int(res['tot_rds']), int(res['n_ambig'])
    if not args.no_quantify_unmapped:
        taxids2abs['Unmapped'][0] += float(n_ambig)
    count, bases, first = res['count'], res['bases'], res['first_seen']
    hit = np.nonzero(count)[0]
    for t in hit[np.argsort(first[hit], kind='stable')]:  # dict order = order of first unique hit (:236-240)
        taxid = taxids[int(t)]
        nreads, nbases = int(count[t]), int(bases[t])
        if args.length_normalize:
            nbases = nbases / taxid2info[taxid][0]  # reference: sum of per-read hitlen/len (:233-234), <=1e-15 rel. apart
        if taxid in taxids2abs:
            taxids2abs[taxid][0] += nreads
            taxids2abs[taxid][1] += nbases
        else:
            taxids2abs[taxid] = [nreads, nbases] + taxid2info[taxid]
    if args.low_mem and (res['mm_nreads'] if 'resident' in res else len(res['mm_hitlen'])) > 0:
        raise TypeError("object of type 'int' has no len()")  # what the reference does at :253,255
    multimapped = multimapped_lists(res, taxids) if want_lists else res
    if not args.no_quantify_unmapped:
        if tot_rds == 0:
            sys.exit('No reads mapped. Aborting...')
        taxids2abs['Unmapped'][1] = taxids2abs['Unmapped'][0] / float(tot_rds)
    return taxids2abs, multimapped, {}


def map_and_process_file(args, path, acc2info, taxid2info, _want_lists=True, _resident=False, _paf=False):
    """map_and_process for a plain SAM FILE, without the text or the records ever being host arrays: the file goes up
    through page-locked chunks (Hip.upload_file), is tokenised where it lands and stage C runs on the records the
    tokeniser left in HBM.  A line the reference cannot parse makes this return None: the caller then takes the
    streaming path, which reproduces the reference's exception for that line."""
    acc_index, taxids, ref2tax = dense_tables(acc2info, taxid2info)
    _ = taxid2info['Unmapped']  # KeyError here, as at :197, when db_info lacks the Unmapped row
    hip = _hip.Hip.get()
    names = [None] * len(acc_index)
    for a, i in acc_index.items():
        names[i] = a
    index = hip.acc_index(names)
    d_text = batch = None
    try:
        # the text plus ~60 B per line of line index and tokeniser output must fit beside what is resident already:
        # otherwise straight to the streaming path (256 MB chunks), without first filling the device and failing
        free, _, pooled = hip.mem_info()
        streamed = os.environ.get('MG_NO_STREAM') != '1'
        # (streamed: only the 16-byte records — ~1/20 of the text — and three chunks of text are ever resident)
        if (os.path.getsize(path) // 8 if streamed else 2 * os.path.getsize(path)) > free + pooled:
            return None
        try:
            if not streamed:  # the whole text up, then one tokeniser call (round 2's path)
                d_text, size = hip.upload_file(path)
                batch = hip.sam_tokenize_dev_batch(d_text.ptr, size, index, '', paf=_paf)
            else:
                # reader threads -> page-locked chunks -> HBM -> tokeniser, chunk i + 1 in flight while chunk i is tokenised
                batch = hip.sam_stream_file(path, index, paf=_paf, chunk_bytes=int(os.environ.get('MG_STREAM_CHUNK_BYTES', 0)))
        except _hip.SamParseError:
            return None
        except _hip.HipError as e:
            # the whole file as ONE batch did not fit (hipMalloc failed, or the text exceeds what one ingest call
            # takes): the streaming path tokenises it in 256 MB chunks instead
            if e.code in (_hip.ERR_NOMEM, _hip.ERR_ARG, _hip.ERR_CAPACITY):  # (capacity: a line longer than a chunk's headroom)
                return None
            raise
        finally:
            if d_text is not None:
                d_text.free()
        res = hip.profile_assign_dev_records(batch.ptr, batch.count, ref2tax, len(taxids), float(args.pct_id), [batch],
                                             resident=_resident)
        batch = None  # owned by the result now
    finally:
        index.free()
        if batch is not None:
            batch.free()
    if not _want_lists:
        res = dict(res, taxids=taxids)
    return assemble_taxids2abs(args, res, taxids, taxid2info, want_lists=_want_lists)


# ---- one process per GPU (python -m torch.distributed.run ... -m metalign_amd.map_and_profile ...) ---------------------
# What costs time in this stage is getting tens of gigabytes of SAM text tokenised; stage C itself takes milliseconds
# for 10^8 records.  So the ranks share the TEXT — every rank uploads and tokenises a line-aligned byte range of the
# file on its GPU — and the 16-byte records (1/20 of the text) are gathered on rank 0, which runs stage C and the tail
# exactly as a single process does, on exactly the record stream a single process sees: the only thing a rank cannot
# know alone, whether its first retained line continues the previous range's last read, is settled from the first /
# last QNAMEs the ranks exchange.

def sam_range_of_rank(path, rank, world):
    """(start, end) byte offsets of the lines that START in rank `rank`'s share of the file."""
    size = os.path.getsize(path)

    def line_start(pos):
        if pos <= 0:
            return 0
        if pos >= size:
            return size
        with open(path, 'rb') as fh:
            p = pos - 1  # (a newline at pos - 1 makes pos itself a line start)
            while p < size:
                fh.seek(p)
                buf = fh.read(1 << 20)
                if not buf:
                    return size
                at = buf.find(b'\n')
                if at >= 0:
                    return min(p + at + 1, size)
                p += len(buf)
        return size
    return line_start(size * rank // world), line_start(size * (rank + 1) // world)


def first_retained_qname(path, start, end):
    """QNAME of the first line in [start, end) that the line filter of :201-217 keeps (None when there is none)."""
    with open(path, 'rb') as fh:
        fh.seek(start)
        left, carry = end - start, b''
        while left > 0:
            buf = fh.read(min(1 << 20, left))
            if not buf:
                break
            left -= len(buf)
            lines = (carry + buf).split(b'\n')
            carry = lines.pop() if left > 0 else b''
            for ln in lines:
                if ln.startswith(b'@'):
                    continue
                f = ln.strip().split()
                if len(f) < 6:
                    continue
                try:
                    flag = int(f[1])
                except ValueError:
                    return None  # (the tokeniser reports the line)
                if (flag & 4) or f[5] == b'*':
                    continue
                return f[0].decode('utf-8', 'replace')
    return None


def clear_continued_heads(first_word_of, counts, firsts, lasts):
    """The pieces of one record stream, tokenised independently (every piece's first record has its new-read bit set):
    for every non-empty piece whose first retained QNAME equals the last retained QNAME in front of it, call
    first_word_of(record index) to clear that bit.  -> the record offsets of the pieces."""
    offs, prev = [0], None
    for n, fq, lq in zip(counts, firsts, lasts):
        if n and prev is not None and fq == prev:
            first_word_of(offs[-1])
        if n:
            prev = lq
        offs.append(offs[-1] + n)
    return offs


def gather_record_pieces(torch, dist, rank, world, mine, n, first_q, last_q, bad, device):
    """The exchange of map_and_process_file_dist, on whatever device the process group runs (cuda under RCCL, cpu under
    gloo in the tests).  mine: this rank's n records as 4 n int32 words (None when n == 0).  Rank 0 gets
    (all records in rank order with the new-read bits at the cuts settled, total); the others 'done'; None (rank 0) when
    some rank could not tokenise its piece."""
    # every rank learns every piece's size, state and boundary names (two small all-gathers)
    word = torch.tensor([n, bad], dtype=torch.int64, device=device)
    words = [torch.zeros_like(word) for _ in range(world)]
    dist.all_gather(words, word)
    counts = [int(w[0].item()) for w in words]
    if any(int(w[1].item()) for w in words):
        return None if rank == 0 else 'done'
    nm = torch.zeros(2, 512, dtype=torch.uint8)
    for row, q in enumerate((first_q, last_q)):
        b = q.encode('utf-8')[:511]
        if b:
            nm[row, :len(b)] = torch.frombuffer(bytearray(b), dtype=torch.uint8)
    nm = nm.to(device)
    nms = [torch.zeros_like(nm) for _ in range(world)]
    dist.all_gather(nms, nm)
    as_str = lambda t: bytes(t.cpu().numpy().tobytes()).split(b'\0', 1)[0].decode('utf-8', 'replace')
    firsts, lasts = [as_str(t[0]) for t in nms], [as_str(t[1]) for t in nms]
    if rank != 0:
        if n:
            dist.send(mine, dst=0)
        return 'done'
    total = sum(counts)
    buf = torch.zeros(max(4 * total, 4), dtype=torch.int32, device=device)
    offs = clear_continued_heads(lambda i: None, counts, firsts, lasts)  # (offsets first; the bits after the pieces are in)
    if n:
        buf[: 4 * n] = mine
    for r in range(1, world):
        if counts[r]:
            dist.recv(buf[4 * offs[r]: 4 * offs[r + 1]], src=r)

    def clear(i):
        buf[4 * i] &= 0x7FFFFFFF  # ref_new is the record's first word; NEW is its top bit
    clear_continued_heads(clear, counts, firsts, lasts)
    return buf, total


class _TensorOwner:
    def __init__(self, t):
        self.t = t

    def free(self):
        self.t = None


def map_and_process_file_dist(args, path, acc2info, taxid2info, ctx, _want_lists=True, _resident=False):
    """map_and_process_file with one process per GPU.  Rank 0 returns what map_and_process_file returns (None: a line
    the reference cannot parse somewhere — it then takes the streaming path over the whole file, which reproduces the
    reference's exception); the other ranks return 'done'."""
    import torch
    dist, rank, world, hip = ctx
    acc_index, taxids, ref2tax = dense_tables(acc2info, taxid2info)
    _ = taxid2info['Unmapped']
    names = [None] * len(acc_index)
    for a, i in acc_index.items():
        names[i] = a
    index = hip.acc_index(names)
    start, end = sam_range_of_rank(path, rank, world)
    d_text = batch = None
    bad = 0
    try:
        try:
            if end > start:
                batch = hip.sam_stream_file(path, index, offset=start, length=end - start)
        except (_hip.SamParseError, _hip.HipError):
            bad = 1
        finally:
            if d_text is not None:
                d_text.free()
        n = batch.count if batch is not None else 0
        fq = (first_retained_qname(path, start, end) or '') if n else ''
        lq = batch.last_qname if n else ''
        dev = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'  # (gloo — tests: staged through the host)
        mine = torch.as_tensor(_CudaWords(batch.ptr, 4 * n), device='cuda') if n else None
        if mine is not None and dev == 'cpu':
            mine = mine.cpu()
        got = gather_record_pieces(torch, dist, rank, world, mine, n, fq, lq, bad, dev)
        if got is None or got == 'done':
            if rank != 0 and n:
                torch.cuda.current_stream().synchronize()  # (the records are freed on the way out)
            return got
        buf, total = got
        if dev == 'cpu':
            d_all = hip.array(buf[: max(4 * total, 4)].numpy())
            ptr, owner = d_all.ptr, d_all
        else:
            torch.cuda.current_stream().synchronize()
            ptr, owner = buf.data_ptr(), _TensorOwner(buf)
        res = hip.profile_assign_dev_records(ptr, total, ref2tax, len(taxids), float(args.pct_id), [owner],
                                             resident=_resident)
    finally:
        index.free()
        if batch is not None:
            batch.free()
    if not _want_lists:
        res = dict(res, taxids=taxids)
    return assemble_taxids2abs(args, res, taxids, taxid2info, want_lists=_want_lists)


class _CudaWords:
    """n int32 words of device memory, for torch.as_tensor (zero-copy)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {'shape': (int(n),), 'typestr': '<i4', 'data': (int(ptr), False), 'version': 2}


def map_and_process(args, instream, acc2info, taxid2info, _assign=None, _want_lists=True, _resident=False):
    """Reference signature (:193).  `_assign` is a test seam; product code never passes it."""
    acc_index, taxids, ref2tax = dense_tables(acc2info, taxid2info)
    _ = taxid2info['Unmapped']  # KeyError here, as at :197, when db_info lacks the Unmapped row
    # test seam: an injected record-level backend is fed by the host tokeniser; product code never passes one
    tokenise = _device_tokenise if _assign is None else tokenise_sam
    if getattr(args, 'paf_input', False):
        tokenise = _device_tokenise_paf if _assign is None else tokenise_paf
    recs = tokenise(instream, acc_index, decode=(args.input_type != 'sam'))
    res = (_assign or (_device_assign_resident if _resident else _device_assign))(recs, ref2tax, len(taxids),
                                                                                  float(args.pct_id))
    if not _want_lists:
        res = dict(res, taxids=taxids)
    return assemble_taxids2abs(args, res, taxids, taxid2info, want_lists=_want_lists)


def preprocess_multimapped(args, multimapped, taxids2abs):
    """Drop hits to taxa without unique reads; drop reads left empty (:180-188)."""
    out = []
    for read in multimapped:
        keep = [t for t in read[:-1] if t in taxids2abs]
        if keep:
            keep.append(read[-1])
            out.append(keep)
    return out


def resolve_multi_prop(args, taxids2abs, multimapped, low_mem_mmap, taxid2info):
    """Split each multimapped read's bases in proportion to unique bases (:269-312)."""
    echo('Assigning multimapped reads...', args.verbose)
    if args.low_mem:
        total = float(sum(v[1] for v in taxids2abs.values()))
        for taxid, nhits in low_mem_mmap.items():
            if taxid not in taxids2abs:
                continue
            share = nhits * (taxids2abs[taxid][1] / total)
            if args.length_normalize:
                share /= taxid2info[taxid][0]
            taxids2abs[taxid][1] += share
        return taxids2abs
    pending = {}
    for read in multimapped:
        taxa = list(dict.fromkeys(t for t in read[:-1] if t in taxids2abs))
        if not taxa:
            continue
        weights = [taxids2abs[t][1] for t in taxa]
        denom = sum(weights)
        if denom == 0.0:
            continue
        hitlen = read[-1]
        for t, w in zip(taxa, weights):
            part = (w / denom) * hitlen
            if args.length_normalize:
                part /= taxid2info[t][0]
            pending[t] = pending[t] + part if t in pending else part
    for t, extra in pending.items():
        taxids2abs[t][1] += extra
    return taxids2abs


def resolve_multi_prop_csr(args, taxids2abs, mm, taxid2info):
    """resolve_multi_prop (:269-312) straight from the kernel's multimapped CSR, by the library's host routine
    (mg_multimapped_shares: 12 ms for the 5M entries of BASELINE configs[2] where the numpy version below takes 80).

    Same arithmetic in the same order as the list version: per read, the DISTINCT taxa that still have an entry in
    taxids2abs share the read's hitlen in proportion to their current bases; every taxon's additions are summed
    in read order and applied once at the end."""
    echo('Assigning multimapped reads...', args.verbose)
    taxids = mm['taxids']
    if len(mm['mm_hitlen']) == 0:
        return taxids2abs
    weight = np.full(len(taxids), np.nan)
    index = {t: i for i, t in enumerate(taxids)}
    for taxid, row in taxids2abs.items():
        weight[index[taxid]] = row[1]
    glen = np.array([taxid2info[x][0] for x in taxids], dtype=np.float64) if args.length_normalize else None
    extra, touched = _hip.multimapped_shares(mm['mm_offsets'], mm['mm_tax'], mm['mm_hitlen'], weight, glen)
    for i in np.nonzero(touched)[0]:
        taxids2abs[taxids[int(i)]][1] += float(extra[i])
    return taxids2abs


def resolve_multi_prop_csr_numpy(args, taxids2abs, mm, taxid2info):
    """The same, vectorised in numpy (np.bincount adds sequentially): what the library routine is checked against."""
    taxids = mm['taxids']
    T = len(taxids)
    off = mm['mm_offsets'].astype(np.int64)
    tax = mm['mm_tax'].astype(np.int64)
    hitlen = mm['mm_hitlen'].astype(np.float64)
    if len(hitlen) == 0:
        return taxids2abs
    weight = np.full(T, np.nan)
    index = {t: i for i, t in enumerate(taxids)}
    for taxid, row in taxids2abs.items():
        weight[index[taxid]] = row[1]
    rid = np.repeat(np.arange(len(hitlen), dtype=np.int64), np.diff(off))
    keep = ~np.isnan(weight[tax])
    pairs = np.unique(rid[keep] * T + tax[keep])  # distinct (read, taxon), ascending by read
    r, t = pairs // T, pairs % T
    w = weight[t]
    denom = np.bincount(r, weights=w, minlength=len(hitlen))
    ok = denom[r] != 0.0
    r, t, w = r[ok], t[ok], w[ok]
    part = (w / denom[r]) * hitlen[r]
    if args.length_normalize:
        part = part / np.array([taxid2info[x][0] for x in taxids], dtype=np.float64)[t]
    extra = np.bincount(t, weights=part, minlength=T)
    touched = np.zeros(T, dtype=bool)
    touched[t] = True
    for i in np.nonzero(touched)[0]:
        taxids2abs[taxids[int(i)]][1] += float(extra[i])
    return taxids2abs


def resolve_multi_prop_device(args, taxids2abs, mm, taxid2info):
    """resolve_multi_prop (:269-312) by mg_profile_resolve_multimapped_dev: the multimapped lists stay on the GPU,
    the host sends one weight per taxon (NaN = no entry in taxids2abs) and gets one addition per taxon back."""
    echo('Assigning multimapped reads...', args.verbose)
    taxids = mm['taxids']
    weight = np.full(len(taxids), np.nan)
    index = {t: i for i, t in enumerate(taxids)}
    for taxid, row in taxids2abs.items():
        weight[index[taxid]] = row[1]
    glen = np.array([taxid2info[x][0] for x in taxids], dtype=np.float64) if args.length_normalize else None
    extra = mm['resident'].resolve_multimapped(weight, glen)
    for i in np.nonzero(extra)[0]:
        taxids2abs[taxids[int(i)]][1] += float(extra[i])
    return taxids2abs


def rank_renormalize(args, clades2abs, only_strains=False):
    """Scale abundances so each rank sums to the mapped percentage (:316-339)."""
    totals = dict.fromkeys(RANKS, 0.0)
    mapped_pct = 100.0
    if not args.no_quantify_unmapped and 'Unmapped' in clades2abs:
        mapped_pct = 100.0 - (100.0 * clades2abs['Unmapped'][-1])
    members = [k for k, v in clades2abs.items()
               if k != 'Unmapped' and not (only_strains and v[1] != 'strain')]
    for k in members:
        totals[clades2abs[k][1]] += clades2abs[k][-1]
    for k in members:
        clades2abs[k][-1] /= (totals[clades2abs[k][1]] / mapped_pct)
    return clades2abs


def gen_lower_taxa(taxids2abs):
    """Push every non-strain taxon down to a synthetic '<taxid>.0 unknown strain' (:344-364)."""
    extra = {}
    for key in taxids2abs:
        taxid, rank, taxlin, namelin, ab = taxids2abs[key]
        if rank == 'strain':
            continue
        label = namelin.split('|')[_RANK_POS[rank]] + ' unknown strain'
        child = taxid + '.0'
        extra[child] = [child, 'strain', taxlin + child, namelin + label, ab]
    taxids2abs.update(extra)
    return {k: v for k, v in taxids2abs.items() if v[1] == 'strain'}


def tree_results_cami(args, taxids2abs):
    """Strain-level profile -> all clades, CAMI field order (:368-399)."""
    for taxid, old in taxids2abs.items():
        taxids2abs[taxid] = [taxid, old[3], old[5], old[4], old[1]]
    taxids2abs = gen_lower_taxa(taxids2abs)
    taxids2abs = rank_renormalize(args, taxids2abs, only_strains=True)
    clades2abs = dict(taxids2abs)  # shallow on purpose: strain rows are shared, as in the reference
    for taxid in taxids2abs:
        tax_path = taxids2abs[taxid][2].split('|')
        name_path = taxids2abs[taxid][3].split('|')
        for depth in range(len(tax_path) - 1):
            clade = tax_path[depth]
            if clade == '':
                continue
            if clade in clades2abs:
                clades2abs[clade][-1] += taxids2abs[taxid][-1]
            else:
                clades2abs[clade] = [clade, RANKS[depth], '|'.join(tax_path[:depth + 1]),
                                     '|'.join(name_path[:depth + 1]), taxids2abs[taxid][-1]]
    if args.rank_renormalize:
        clades2abs = rank_renormalize(args, clades2abs)
    return clades2abs


def compute_abundances(args, infile, acc2info, tax2info):
    """One input file -> clade abundances (:404-433)."""
    if args.input_type == 'sam':
        instream = open(infile, 'rb')  # bytes go to the device tokeniser as they are
    else:  # stream minimap2's SAM, exactly the reference's invocation (:413-416)
        mapper = subprocess.Popen(['minimap2', '-ax', 'sr', '-t', str(args.threads), '-2', '-n' '1',
                                   '--secondary=yes', args.db, infile], stdout=subprocess.PIPE, bufsize=1)
        instream = iter(mapper.stdout.readline, b'')
    # product path: the multimapped reads stay in the kernel's CSR form; preprocess_multimapped (:180-188) is
    # subsumed by the membership test inside the resolve step (a taxon dropped there is dropped here too)
    on_device = bool(getattr(args, 'device_multimap', False))
    done = None
    seams_untouched = (_device_tokenise is tokenise_sam_device and _device_tokenise_paf is tokenise_paf_device
                       and _device_assign is _DEVICE_ASSIGN)  # (tests reroute them)
    paf = bool(getattr(args, 'paf_input', False))
    if args.input_type == 'sam' and seams_untouched and not paf:
        from .select_db import dist_context
        ctx = dist_context()
        if ctx is not None:  # one process per GPU: the ranks tokenise the text between them, rank 0 does the rest
            done = map_and_process_file_dist(args, infile, acc2info, tax2info, ctx, _want_lists=False, _resident=on_device)
            if done == 'done':
                instream.close()
                return None
            seams_untouched = seams_untouched and done is not None  # (None: rank 0 streams the file, alone)
    if args.input_type == 'sam' and seams_untouched and done is None:
        # a plain file (SAM, or a PAF replay): all the way on the device
        done = map_and_process_file(args, infile, acc2info, tax2info, _want_lists=False, _resident=on_device, _paf=paf)
    if done is None:
        done = map_and_process(args, instream, acc2info, tax2info, _want_lists=False, _resident=on_device)
    taxids2abs, mm, low_mem_mmap = done
    if args.input_type == 'sam':
        instream.close()
    else:
        mapper.stdout.close()
        mapper.wait()
    taxids2abs = {k: v for k, v in taxids2abs.items() if v[0] > args.read_cutoff}
    if on_device:
        try:
            if mm['mm_nreads'] > 0:
                taxids2abs = resolve_multi_prop_device(args, taxids2abs, mm, tax2info)
        finally:
            mm['resident'].free()
    elif len(mm['mm_hitlen']) > 0:
        taxids2abs = resolve_multi_prop_csr(args, taxids2abs, mm, tax2info)
    return tree_results_cami(args, taxids2abs)


def gather_results(args, acc2info, taxid2info):
    """Average over input files and bucket by rank (:438-463)."""
    merged = {}
    for infile in args.infiles:
        echo('Computing abundances for input file: ' + infile, args.verbose)
        clades = compute_abundances(args, infile, acc2info, taxid2info)
        if clades is None:  # a rank other than 0 of a multi-GPU launch: its share of the work is done
            args._not_root = True
            continue
        for clade, row in clades.items():
            if clade in merged:
                merged[clade][-1] += row[-1]
            else:
                merged[clade] = row
    merged.pop('Unmapped', None)
    echo('Compiling and writing results...', args.verbose)
    rank_results = {i: [] for i in range(len(RANKS))}
    nfiles = len(args.infiles)
    for row in merged.values():
        row[4] = row[4] / nfiles
        depth = _RANK_POS[row[1]]
        if depth == 7:
            row.extend([row[0], row[0].split('.')[0]])  # _CAMI_genomeID, _CAMI_OTU
        rank_results[depth].append(row)
    return rank_results


def write_results(args, rank_results):
    """CAMI profile (:467-494)."""
    with open(args.output, 'w') as out:
        sample = ','.join(args.infiles) if args.sampleID == 'NONE' else args.sampleID
        out.write('@SampleID:' + sample + '\n')
        out.write('@Version:Metalign\n')
        out.write('@Ranks: ' + '|'.join(RANKS) + '\n\n')
        out.write('\t'.join(['@@TAXID', 'RANK', 'TAXPATH', 'TAXPATHSN', 'PERCENTAGE',
                             '_CAMI_genomeID', '_CAMI_OTU']) + '\n')
        for depth in range(len(RANKS)):
            rows = rank_results[depth]
            rows.sort(key=lambda r: 100.0 - r[4])
            for row in rows:
                if row[4] < args.min_abundance:
                    continue
                row[4] = 0.00001 if row[4] < 0.00001 else float('%.5f' % row[4])
                out.write('\t'.join(str(x) for x in row) + '\n')


def map_main(args=None):
    if args is None:
        args = profile_parseargs()
    if args.pct_id > 1.0 or args.pct_id < 0.0:
        sys.exit('Error: --pct_id must be between 0.0 and 1.0, inclusive.')
    if args.db == 'NONE' and not args.infiles[0].endswith(('sam', 'paf')):
        sys.exit('Error: --db must be specified unless sam files are provided.')
    args.data = cli.with_slash(args.data)
    if args.dbinfo == 'AUTO':
        args.dbinfo = args.data + 'db_info.txt'
    if args.input_type == 'AUTO':
        first = args.infiles[0]
        if first.endswith('.sam'):
            args.input_type = 'sam'
        elif first.endswith('.paf'):  # build-only: a minimap2 PAF file is replayed like a SAM file
            args.input_type, args.paf_input = 'sam', True
        else:
            args.input_type = cli.sniff_reads_type(first)
    if int(os.environ.get('RANK', '0')) != 0 and (args.input_type != 'sam' or getattr(args, 'paf_input', False)):
        return  # a multi-GPU launch shares only SAM files between the ranks; anything else is rank 0's alone
    if int(os.environ.get('RANK', '0')) == 0:
        open(args.output, 'w').close()
    acc2info, taxid2info = get_acc2info(args)
    rank_results = gather_results(args, acc2info, taxid2info)
    if not getattr(args, '_not_root', False):
        write_results(args, rank_results)


if __name__ == '__main__':
    map_main(profile_parseargs())
