"""Data formats either side of the hot path: reads (FASTA / FASTQ, optionally gzipped) in, and the
on-disk genome sketch table that replaces the reference's CMash artefacts.

Sketch table directory (replaces data/cmash_db_n1000_k60.h5, ..._dump.kmc_{pre,suf} and
..._30-60-10.bf, /root/reference/scripts/select_db.py:44,69-70):

    <dir>/meta.json           {"format": "metalign_amd.sketch_table", "version": 1, "n": 1000,
                               "ks": [21, 31, 51], "ngenomes": G, "hash": "murmur3_x64_128.h1(canonical ASCII k-mer)"}
    <dir>/names.txt           one organism file name per line (taxid_<a>_<b>_genomic.fna.gz,
                               /root/reference/utils/ncbi2db.py:160-186), row order == genome id
    <dir>/k<K>.hashes.u64     little-endian u64, every genome's ascending sketch back to back
    <dir>/k<K>.offsets.u64    little-endian u64[G+1]
Flat files so that a RefSeq-scale table (1.6 GB per k) is np.memmap'ed and uploaded shard by shard.
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
    """Host view of a sketch table directory."""

    def __init__(self, path):
        self.path = path
        with open(os.path.join(path, "meta.json")) as fh:
            self.meta = json.load(fh)
        if self.meta.get("format") != TABLE_FORMAT:
            raise ValueError("%s is not a %s directory" % (path, TABLE_FORMAT))
        self.ks = [int(k) for k in self.meta["ks"]]
        self.n = int(self.meta["n"])
        with open(os.path.join(path, "names.txt")) as fh:
            self.names = [ln.rstrip("\n") for ln in fh]
        self.ngenomes = len(self.names)

    def arrays(self, k):
        h = np.memmap(os.path.join(self.path, "k%d.hashes.u64" % k), dtype="<u8", mode="r")
        o = np.fromfile(os.path.join(self.path, "k%d.offsets.u64" % k), dtype="<u8")
        assert len(o) == self.ngenomes + 1 and int(o[-1]) == len(h)
        return h, o


def write_sketch_table(path, names, ks, n, per_k):
    """per_k: {k: (hashes u64[], offsets u64[G+1])}."""
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "names.txt"), "w") as fh:
        for nm in names:
            fh.write(nm + "\n")
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
