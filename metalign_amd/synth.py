"""Seeded synthetic workloads: genomes, 150 bp reads, alignment records.

There is no network for real data sets; bench.py and the full-size property tests use these generators.
PRNG: numpy Generator(PCG64) seeded with (0x4D37A + stream id); all draws are documented below so that
the CPU baseline and the GPU path see the same bytes.
"""
import numpy as np

from . import _hip

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b

SEED = 0x4D37A


def make_genomes(ngenomes, length, seed=SEED):
    """G iid-uniform ACGT sequences of equal length. -> (bases u8[G*length], offsets u64[G+1])."""
    rng = np.random.default_rng(seed)
    bases = _ACGT[rng.integers(0, 4, size=ngenomes * length, dtype=np.uint8)]
    offsets = np.arange(ngenomes + 1, dtype=np.uint64) * np.uint64(length)
    return bases, offsets


def make_reads(gbases, goffsets, nreads, readlen=150, npresent=50, err=0.01, seed=SEED + 1):
    """Reads = substrings of `npresent` genomes under a log-normal abundance vector, both strands,
    `err` substitution rate. -> (bases u8[nreads*readlen], offsets u64[nreads+1], source genome i64[nreads])."""
    rng = np.random.default_rng(seed)
    g = len(goffsets) - 1
    present = rng.choice(g, size=min(npresent, g), replace=False)
    w = rng.lognormal(0.0, 1.0, size=len(present))
    src = present[rng.choice(len(present), size=nreads, p=w / w.sum())]
    glen = (goffsets[1:] - goffsets[:-1]).astype(np.int64)
    start = (rng.random(nreads) * (glen[src] - readlen)).astype(np.int64) + goffsets[src].astype(np.int64)
    rev = rng.random(nreads) < 0.5
    out = np.empty(nreads * readlen, dtype=np.uint8)
    ar = np.arange(readlen, dtype=np.int64)
    step = 200000
    for a in range(0, nreads, step):
        b = min(a + step, nreads)
        idx = start[a:b, None] + np.where(rev[a:b, None], readlen - 1 - ar[None, :], ar[None, :])
        blk = gbases[idx]
        blk[rev[a:b]] = _COMP[blk[rev[a:b]]]
        out[a * readlen: b * readlen] = blk.reshape(-1)
    if err > 0:
        nerr = rng.binomial(out.size, err)
        pos = rng.integers(0, out.size, size=nerr)
        out[pos] = _ACGT[rng.integers(0, 4, size=nerr, dtype=np.uint8)]
    offsets = np.arange(nreads + 1, dtype=np.uint64) * np.uint64(readlen)
    return out, offsets, src


def make_alignment_records(src_ref, nref, readlen=150, seed=SEED + 2):
    """Alignment replay for reads whose true accession row is src_ref[i] (single-end):
    70 % one passing primary line, 25 % primary + one secondary (SEQ '*') to a sibling accession,
    5 % one line failing pct_id 0.5.  L = 1.25 lines / read. -> REC_DTYPE array in read order."""
    rng = np.random.default_rng(seed)
    n = len(src_ref)
    u = rng.random(n)
    has_sec = (u >= 0.70) & (u < 0.95)
    fails = u >= 0.95
    nlines = 1 + has_sec.astype(np.int64)
    first = np.zeros(n + 1, dtype=np.int64)
    first[1:] = np.cumsum(nlines)
    recs = np.zeros(int(first[-1]), dtype=_hip.REC_DTYPE)
    strand = (rng.random(n) < 0.5).astype(np.uint32) * 16
    p = first[:-1]
    recs["ref_new"][p] = src_ref.astype(np.uint32) | np.uint32(_hip.NEW_BIT)
    recs["total"][p] = readlen
    recs["matched"][p] = np.where(fails, rng.integers(10, readlen // 2 - 5, size=n), readlen - rng.integers(0, 8, size=n))
    recs["flag_len"][p] = strand | np.uint32(readlen << _hip.LEN_SHIFT)
    s = p[has_sec] + 1
    sib = (src_ref[has_sec] + rng.integers(1, 4, size=int(has_sec.sum()))) % nref
    sib = np.where(sib == 0, 1, sib)  # row 0 is the 'Unmapped' pseudo-accession
    recs["ref_new"][s] = sib.astype(np.uint32)
    recs["total"][s] = readlen
    recs["matched"][s] = readlen - rng.integers(0, 12, size=len(s))
    recs["flag_len"][s] = 256 | strand[has_sec]
    return recs
