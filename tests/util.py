"""Shared helpers for the parity tests (synthetic reads / genomes; not product code)."""
import numpy as np


def random_genomes(rng, ngenomes, length, with_n=False):
    bases = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=ngenomes * length).astype(np.uint8)
    if with_n:
        idx = rng.integers(0, bases.size, size=max(bases.size // 500, 1))
        bases[idx] = ord("N")
    offsets = (np.arange(ngenomes + 1, dtype=np.uint64) * np.uint64(length))
    return bases, offsets


_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTNacgtn", b"TGCANtgcan"):
    _COMP[a] = b


def sample_reads(rng, gbases, goffsets, nreads, readlen, err=0.01, ragged=False, lower=False, present=None):
    """Reads drawn from genomes (both strands, substitution errors). -> (bases u8, offsets u64, source genome)."""
    g = len(goffsets) - 1
    present = np.arange(g) if present is None else np.asarray(present)
    src = present[rng.integers(0, len(present), size=nreads)]
    lens = np.full(nreads, readlen, dtype=np.int64)
    if ragged:
        lens = rng.integers(max(readlen // 3, 1), readlen + 1, size=nreads)
    glen = (goffsets[1:] - goffsets[:-1]).astype(np.int64)
    start = (rng.random(nreads) * np.maximum(glen[src] - lens, 1)).astype(np.int64) + goffsets[src].astype(np.int64)
    offsets = np.zeros(nreads + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(lens)
    out = np.empty(int(offsets[-1]), dtype=np.uint8)
    rev = rng.random(nreads) < 0.5
    for i in range(nreads):
        s = gbases[start[i]: start[i] + lens[i]]
        if rev[i]:
            s = _COMP[s[::-1]]
        out[int(offsets[i]): int(offsets[i + 1])] = s
    if err > 0:
        m = rng.random(out.size) < err
        out[m] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(m.sum()))
    if lower:
        m = rng.random(out.size) < 0.1
        out[m] = out[m] | 0x20
    return out, offsets, src
