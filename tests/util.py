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


def job_sketch_size(oracle, job, ki, bases, offsets, k, table_hashes, s=0):
    """How many hashes the read sketch of a ShardJob holds for its k number ki: what passes the table's bit filter — or, when
    the job has built the table's resident index (dense tables, s = 0), exactly the read k-mers that are hashes of the table."""
    filt = job.engine.filters[ki] if ki < len(job.engine.filters) else None
    hmax = int(table_hashes.max())
    if s == 0 and filt is not None and filt.resident_bytes > 0:
        h = oracle.sketch_reads(bases, offsets, k, hmax=hmax)[0]
        return int(np.isin(h, table_hashes).sum())
    return len(oracle.sketch_reads_filtered(bases, offsets, k, table_hashes, hmax=hmax, s=s)[0])


def flat(seqs):
    """list of bytes -> (bases u8, offsets u64)"""
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(r) for r in seqs])
    return np.frombuffer(b"".join(seqs), dtype=np.uint8), offs


def refpipe_case(rng, ngenomes=5, glen=(1200, 2000), nreads=60, strains=True):
    """Genomes that SHARE k-mers (a strain = a mutated copy; repeats inside one genome; a reverse-complemented copy), N runs,
    lower case, degenerate genomes; reads from three of them, both strands, each twice so that ci = 2 is met, plus noise.
    -> (genomes: list of bytes, reads: list of bytes)"""
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)

    def rand(n, lo, hi, p_n=0.002, p_lower=0.05):
        out = []
        for _ in range(n):
            ln = int(rng.integers(lo, hi + 1))
            b = rng.choice(alpha, size=ln).astype(np.uint8)
            b[rng.random(ln) < p_n] = ord("N")
            m = rng.random(ln) < p_lower
            b[m] |= 0x20
            out.append(b.tobytes())
        return out
    genomes = rand(ngenomes, *glen)
    if strains:
        g0 = bytearray(genomes[0].upper())
        for p in rng.integers(0, len(g0), size=12):
            g0[int(p)] = b"ACGT"[int(rng.integers(0, 4))]
        genomes.append(bytes(g0))
        genomes.append(genomes[1][:600] + genomes[1][100:700])
        genomes.append(genomes[2].upper().translate(bytes.maketrans(b"ACGT", b"TGCA"))[::-1])
    genomes += [b"ACGT" * 5, b"", b"A" * 300]
    comp = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")
    reads = []
    for g in (0, 2, 4):
        src = genomes[g]
        for _ in range(nreads):
            a = int(rng.integers(0, max(len(src) - 150, 1)))
            r = src[a:a + 150]
            if rng.random() < 0.5:
                r = r.translate(comp)[::-1]
            reads += [r, r]
    reads += rand(40, 60, 150, p_n=0.01, p_lower=0.1) + [b"", b"ACG"]
    return genomes, reads
