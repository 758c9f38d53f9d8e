"""A SECOND, independent restatement of stage A / A' / B (test infrastructure only).

oracle/mg_oracle.c is the normative statement of the sketch / containment arithmetic (KMC 3 and CMash are not
vendored by the reference: /root/reference/.gitignore:5-13), which made one C file the only
definition of truth for the dominant kernel.  This module states the same definitions again, from the prose of
DESIGN.md §2 and NOT from the C source, in a deliberately different style so that the two can only agree by
both being right:

  * MurmurHash3_x64_128 on Python integers (masking to 64 bits), reading the key with int.from_bytes;
  * sequences split into maximal [ACGTacgt] stretches by a regular expression, upper-cased as strings;
  * canonical k-mer = min(kmer, reverse_complement(kmer)) on Python strings (str.translate + slicing);
  * occurrence counting in a dict, saturating at `cs` (kmc -cs<cs>, /root/reference/scripts/select_db.py:50);
  * containment by set membership.

tests/test_oracle_independent.py checks the C oracle against this on seeded inputs (CPU suite).
"""
import re

M64 = (1 << 64) - 1
_C1 = 0x87C37B91114253D5
_C2 = 0x4CF5AD432745937F
RESERVED = M64  # never a sketch member

_RUNS = re.compile(rb"[ACGTacgt]+")
_COMP = bytes.maketrans(b"ACGT", b"TGCA")


def _rotl(x, r):
    return ((x << r) | (x >> (64 - r))) & M64


def _fmix(v):
    v ^= v >> 33
    v = (v * 0xFF51AFD7ED558CCD) & M64
    v ^= v >> 33
    v = (v * 0xC4CEB9FE1A85EC53) & M64
    v ^= v >> 33
    return v


def murmur3_x64_128(key: bytes, seed: int = 0):
    """(h1, h2) of Appleby's MurmurHash3_x64_128, from the published description of the algorithm."""
    h1 = h2 = seed & 0xFFFFFFFF
    n = len(key)
    full = n - (n % 16)
    for off in range(0, full, 16):
        k1 = int.from_bytes(key[off:off + 8], "little")
        k2 = int.from_bytes(key[off + 8:off + 16], "little")
        k1 = (_rotl((k1 * _C1) & M64, 31) * _C2) & M64
        h1 ^= k1
        h1 = (_rotl(h1, 27) + h2) & M64
        h1 = (h1 * 5 + 0x52DCE729) & M64
        k2 = (_rotl((k2 * _C2) & M64, 33) * _C1) & M64
        h2 ^= k2
        h2 = (_rotl(h2, 31) + h1) & M64
        h2 = (h2 * 5 + 0x38495AB5) & M64
    tail = key[full:]
    if len(tail) > 8:
        k2 = int.from_bytes(tail[8:], "little")
        h2 ^= (_rotl((k2 * _C2) & M64, 33) * _C1) & M64
    if tail:
        k1 = int.from_bytes(tail[:8], "little")
        h1 ^= (_rotl((k1 * _C1) & M64, 31) * _C2) & M64
    h1 ^= n
    h2 ^= n
    h1 = (h1 + h2) & M64
    h2 = (h2 + h1) & M64
    h1, h2 = _fmix(h1), _fmix(h2)
    h1 = (h1 + h2) & M64
    h2 = (h2 + h1) & M64
    return h1, h2


CMASH_PRIME = 9999999999971  # CMash: get_prime_lt_x(9999999999971.) — the number is itself prime
HASH_MODE = 0  # 0: hash(min(kmer, revcomp)), 64 bits; 1: min(hash(kmer), hash(revcomp)) % CMASH_PRIME


def canonical_kmers(seq: bytes, k: int):
    """Every k-mer of `seq` (upper case) in order of occurrence, as what kmer_hash takes: the lexicographically smaller
    strand (mode 0) or the pair of strands (mode 1); windows never span a non-ACGT symbol."""
    for m in _RUNS.finditer(seq):
        run = m.group().upper()
        for i in range(len(run) - k + 1):
            kmer = run[i:i + k]
            rc = kmer.translate(_COMP)[::-1]
            if HASH_MODE == 1:
                yield (kmer, rc)
            else:
                yield kmer if kmer <= rc else rc


def kmer_hash(kmer) -> int:
    if isinstance(kmer, tuple):  # mode 1: both strands hashed, the smaller value kept, modulo the prime
        return min(murmur3_x64_128(kmer[0], 0)[0], murmur3_x64_128(kmer[1], 0)[0]) % CMASH_PRIME
    return murmur3_x64_128(kmer, 0)[0]


def sketch_reads(reads, k, hmax=M64, s=0, cs=0, member=None):
    """reads: iterable of bytes.  -> (ascending [(hash, count)], truncated, kmers_seen).
    count = occurrences, saturating at cs when cs > 0; member: optional predicate (the membership pre-filter)."""
    occ, seen = {}, 0
    for r in reads:
        for km in canonical_kmers(r, k):
            seen += 1
            h = kmer_hash(km)
            if h > hmax or h == RESERVED:
                continue
            if member is not None and not member(h):
                continue
            occ[h] = occ.get(h, 0) + 1
    items = sorted(occ.items())
    truncated = bool(s) and len(items) > s
    if truncated:
        items = items[:s]
    if cs:
        items = [(h, min(c, cs)) for h, c in items]
    return items, truncated, seen


def sketch_genome(seq: bytes, k: int, n: int):
    """Bottom-n distinct canonical k-mer hashes of one genome, ascending."""
    hs = {kmer_hash(km) for km in canonical_kmers(seq, k)}
    hs.discard(RESERVED)
    return sorted(hs)[:n]


def sketch_genome_prefix(seq: bytes, kmax: int, k: int, n: int):
    """The k < kmax table entry of one genome under hash mode 1: sketch the genome at kmax keeping, per hash
    value, the k-mer as CMash's CountEstimator.add keeps it (the strand with the smaller hash, the reverse complement on a
    tie); the table holds the distinct mode-1 hashes of the first k bases of those k-mers.  Written from the prose, on Python
    strings and dicts; independent of HASH_MODE."""
    best = {}  # hash value -> smallest prefix key among the kmax-mers that have it
    for m in _RUNS.finditer(seq):
        run = m.group().upper()
        for i in range(len(run) - kmax + 1):
            kmer = run[i:i + kmax]
            rc = kmer.translate(_COMP)[::-1]
            hf, hr = murmur3_x64_128(kmer, 0)[0], murmur3_x64_128(rc, 0)[0]
            kept = rc if hr <= hf else kmer
            pre = kept[:k]
            prc = pre.translate(_COMP)[::-1]
            key = min(murmur3_x64_128(pre, 0)[0], murmur3_x64_128(prc, 0)[0]) % CMASH_PRIME
            h = min(hf, hr) % CMASH_PRIME
            if h not in best or key < best[h]:
                best[h] = key
    chosen = [best[h] for h in sorted(best)[:n]]
    return sorted(set(chosen))


def containment(sketch_items, truncated, ci, genome_sketches):
    """-> [(hits, size)] per genome: size = genome hashes <= bound, hits = those present in the sample at count >= ci."""
    bound = sketch_items[-1][0] if (truncated and sketch_items) else M64
    present = {h for h, c in sketch_items if c >= ci}
    out = []
    for g in genome_sketches:
        inb = [h for h in g if h <= bound]
        out.append((sum(1 for h in inb if h in present), len(inb)))
    return out


# ---- the reference's own wiring of stage A/B ("the reference pipeline"), on STRINGS ---------------------------
# Written from /root/reference/scripts/select_db.py:43-76 and the prose, not from oracle/mg_oracle.c: KMC counts the reads'
# k_max-mers (canonical = the lexicographically smaller strand, count >= ci), kmc_tools intersects them with the k_max-mers of
# the genome sketches, and the streaming query looks the k-prefixes of every surviving k_max-mer AND of its reverse complement up
# among the sketched k_max-mers (kept in the orientation the sketch stored them).  No hash on the query side at all: k-mers are
# matched as strings, prefixes live in Python sets.  (The oracle matches k_max-mers by their hash value; the two can differ only
# by a hash collision.)

def _revcomp(s: bytes) -> bytes:
    return s.translate(_COMP)[::-1]


def refpipe_genome_kmers_forward(seq: bytes, kmax: int, n: int):
    """`build_db --sketch_hash forward`: the kmax-mers of one genome with the n smallest distinct MurmurHash3(k-mer as it stands) %
    9999999999971, per value the first window, kept AS THEY STAND (CMash's training without reverse complements, as recollected).
    A genome that holds a k-mer and its reverse complement may list both: two strings of the sketch."""
    first = {}
    for m in _RUNS.finditer(seq):
        run = m.group().upper()
        for i in range(len(run) - kmax + 1):
            kmer = run[i:i + kmax]
            h = murmur3_x64_128(kmer, 0)[0] % CMASH_PRIME
            if h not in first:
                first[h] = kmer
    return [first[h] for h in sorted(first)[:n]]


def refpipe_genome_kmers(seq: bytes, kmax: int, n: int):
    """The sketched kmax-mers of one genome as the table keeps them: bottom-n by hash (the definition HASH_MODE selects), per hash
    the first window that has it, oriented as the sketch stores it — mode 0: the lexicographically smaller strand; mode 1: the
    strand with the smaller MurmurHash3, the reverse complement on a tie (CMash's CountEstimator.add, as recollected)."""
    first = {}
    for m in _RUNS.finditer(seq):
        run = m.group().upper()
        for i in range(len(run) - kmax + 1):
            kmer = run[i:i + kmax]
            rc = _revcomp(kmer)
            if HASH_MODE == 1:
                hf, hr = murmur3_x64_128(kmer, 0)[0], murmur3_x64_128(rc, 0)[0]
                h, kept = min(hf, hr) % CMASH_PRIME, (rc if hr <= hf else kmer)
            else:
                kept = kmer if kmer <= rc else rc
                h = murmur3_x64_128(kept, 0)[0]
            if h != RESERVED and h not in first:
                first[h] = kept
    return [first[h] for h in sorted(first)[:n]]


def refpipe_query(reads, genome_kmers, ks, ci):
    """reads: iterable of bytes; genome_kmers: per genome the list refpipe_genome_kmers returned; ks ascending, ks[-1] = kmax.
    -> per k a list of (hits, size) per genome."""
    kmax = ks[-1]
    counts = {}
    for r in reads:  # kmc -k<kmax> -ci<ci>: canonical k-mers of the reads with their occurrence counts (:50-52)
        for m in _RUNS.finditer(r):
            run = m.group().upper()
            for i in range(len(run) - kmax + 1):
                kmer = run[i:i + kmax]
                rc = _revcomp(kmer)
                c = kmer if kmer <= rc else rc
                counts[c] = counts.get(c, 0) + 1
    db = set()  # the KMC database of the sketches' k-mers (local_tests/retrain_and_test_metalign.sh:59-66): canonical forms
    for g in genome_kmers:
        for y in g:
            db.add(min(y, _revcomp(y)))
    survivors = [x for x, c in counts.items() if c >= ci and x in db]  # kmc_tools simple ... intersect (:54-56) -> the FASTA (:58-65)
    out = []
    for k in ks:  # the streaming query, k range (:73-76): both strands of every query k-mer, k-prefixes against the tree
        seen = set()
        for x in survivors:
            seen.add(x[:k])
            seen.add(_revcomp(x)[:k])
        col = []
        for g in genome_kmers:
            prefixes = {y[:k] for y in g}
            col.append((len(prefixes & seen), len(prefixes)))
        out.append(col)
    return out
