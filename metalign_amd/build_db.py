#! /usr/bin/env python
"""Build the genome sketch table the pre-filter consumes (metalign_amd/formats.py).

Counterpart of the reference's offline recipe — CMash MakeStreamingDNADatabase.py -n 1000 -k 60, the bloom
pre-filter and the KMC dump (/root/reference/local_tests/retrain_and_test_metalign.sh:49-66) — as one GPU
pass per k (mg_sketch_genomes: k_hash_positions + segmented sort + k_take_bottom_n).

    python -m metalign_amd.build_db <organism_dir | file-list.txt> <out_dir> [-n 1000] [-k 30,40,50,60]
"""
import argparse
import os

import numpy as np

from . import _hip, formats


def genome_bases(path):
    """All records of one organism file as one byte string; records are joined by 'N' so that no k-mer
    spans two contigs."""
    kind = 'fasta'
    b, o, _ = formats.read_sequences(path, kind)
    if len(o) <= 2:
        return b
    parts = []
    for i in range(len(o) - 1):
        if i:
            parts.append(np.frombuffer(b'N', dtype=np.uint8))
        parts.append(b[int(o[i]):int(o[i + 1])])
    return np.concatenate(parts)


def build_reference_pipeline(paths, out_dir, ks, n, batch_bases=1 << 27, hash_mode=0, sketch_hash='canonical'):
    """The table of the REFERENCE PIPELINE (formats.py, version 3): the genomes are sketched at the LARGEST k only, with their
    k-mers kept (CMash: MakeStreamingDNADatabase.py -n 1000 -k 60, /root/reference/local_tests/retrain_and_test_metalign.sh:49),
    and what derives the smaller k's columns from the matched k_max-mers is prepared on the device (mg_refdb_build) — the role
    of the prefix tree inside CMash's database and of the KMC dump of the sketches' k-mers (:59-66).
    sketch_hash = 'forward': a genome's entries are SELECTED by MurmurHash3(k-mer as it stands) % 9999999999971 and kept as they
    stand — CMash's training without reverse complements, as recollected (unverified: DESIGN.md §2); what an entry matches by is
    unchanged (hash_mode), so the query side is the same."""
    hip = _hip.Hip.get()
    previous = hip.hash_mode
    hip.set_hash_mode(hash_mode)
    try:
        ks = sorted(int(k) for k in ks)
        names = [os.path.basename(p) for p in paths]
        hs, his, los, offs = [], [], [], [0]
        i = 0
        while i < len(paths):
            seqs, total = [], 0
            while i < len(paths) and (not seqs or total < batch_bases):
                g = genome_bases(paths[i])
                seqs.append(g)
                total += len(g)
                i += 1
            o = np.zeros(len(seqs) + 1, dtype=np.uint64)
            o[1:] = np.cumsum([len(s) for s in seqs])
            bases = np.concatenate(seqs) if total else np.zeros(1, np.uint8)
            h, hi, lo, go = hip.sketch_genomes_kmers(bases, o, ks[-1], n, sketch_hash=sketch_hash)
            hs.append(h)
            his.append(hi)
            los.append(lo)
            base = offs[-1]
            offs.extend(int(v) + base for v in go[1:])

        def cat(parts):
            return np.concatenate(parts) if parts else np.zeros(0, np.uint64)
        table = hip.refdb_build(cat(hs), cat(his), cat(los), np.asarray(offs, dtype=np.uint64), ks)
        try:
            arrays = table.download()
        finally:
            table.free()
        f = hip.filter_build(arrays["pair_hash"])
        bits = f.download()
        f.free()
        formats.write_refpipe_table(out_dir, names, n, arrays, bits, hash_mode=hash_mode, sketch_hash=sketch_hash)
        return arrays
    finally:
        hip.set_hash_mode(previous)


def build(paths, out_dir, ks, n, batch_bases=1 << 27, hash_mode=0, prefix_tables=False):
    """hash_mode: 0 = MurmurHash3 of the canonical k-mer (default); 1 = min(hash(kmer), hash(revcomp)) % 9999999999971,
    CMash's CountEstimator as SURVEY.md §8(c) recollects it (unverified; include/metalign_hip.h: mg_set_hash_mode).  The
    table records its mode and select_db sketches the reads in the same one.  prefix_tables (mode 1 only): the tables of the
    k below the largest hold the k-PREFIXES of the sketched k_max-mers (CMash's smaller-k columns as recollected), not
    sketches of their own; select_db then runs those k without a hash threshold (the table's largest key is near the
    prime), the stored membership filter doing the rejecting."""
    hip = _hip.Hip.get()
    previous = hip.hash_mode
    hip.set_hash_mode(hash_mode)
    try:
        return _build(hip, paths, out_dir, ks, n, batch_bases, hash_mode, prefix_tables)
    finally:
        hip.set_hash_mode(previous)


def _build(hip, paths, out_dir, ks, n, batch_bases, hash_mode, prefix_tables):
    names = [os.path.basename(p) for p in paths]
    per_k = {k: ([], [0]) for k in ks}
    i = 0
    while i < len(paths):
        seqs, total = [], 0
        while i < len(paths) and (not seqs or total < batch_bases):
            g = genome_bases(paths[i])
            seqs.append(g)
            total += len(g)
            i += 1
        offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
        offs[1:] = np.cumsum([len(s) for s in seqs])
        bases = np.concatenate(seqs) if total else np.zeros(1, np.uint8)
        for k in ks:
            # hash mode 1 with prefix tables: the k < k_max tables hold the k-prefixes of the sketched k_max-mers
            if hash_mode == 1 and prefix_tables and k < max(ks):
                h, o = hip.sketch_genomes_prefix(bases, offs, max(ks), k, n)
            else:
                h, o = hip.sketch_genomes(bases, offs, k, n)
            hs, os_ = per_k[k]
            base = os_[-1]
            hs.append(h)
            os_.extend(int(v) + base for v in o[1:])
    final = {k: (np.concatenate(per_k[k][0]) if per_k[k][0] else np.zeros(0, np.uint64),
                 np.asarray(per_k[k][1], dtype=np.uint64)) for k in ks}
    filters = {}
    for k in ks:  # the membership pre-filter of every k, stored next to the table (the reference's ...bf file)
        f = hip.filter_build(final[k][0])
        filters[k] = f.download()
        f.free()
    formats.write_sketch_table(out_dir, names, ks, n, final, filters, hash_mode=hash_mode, prefix_tables=bool(prefix_tables and hash_mode == 1))
    return final


def main(argv=None):
    p = argparse.ArgumentParser(description='Build the MI355X genome sketch table from organism FASTA files.')
    p.add_argument('genomes', help='Directory of organism files (taxid_*_genomic.fna[.gz]) or a text file listing them.')
    p.add_argument('out_dir', help='Sketch table directory to write (default location: data/sketch_table).')
    p.add_argument('-n', '--num_hashes', type=int, default=1000, help='Sketch size per genome. Default: 1000')
    p.add_argument('-k', '--ks', default='30,40,50,60', help='Comma-separated k-mer sizes. Default: 30,40,50,60')
    p.add_argument('--hash_mode', choices=['canonical', 'cmash'], default='canonical',
                   help="canonical: MurmurHash3 of the lexicographically smaller strand, 64 bits (default). cmash: "
                        "min(hash(kmer), hash(revcomp)) %% 9999999999971, CMash's definition as recollected (unverified).")
    p.add_argument('--prefix_tables', action='store_true',
                   help="with --hash_mode cmash: the k < k_max tables hold the k-prefixes of the sketched k_max-mers (CMash's "
                        "smaller-k columns as recollected) instead of a sketch per k.")
    p.add_argument('--reference_pipeline', action='store_true',
                   help="Build the table for stage A/B wired as the reference wires KMC and CMash (select_db.py:50-59,73-76): reads are "
                        "sketched at the largest k only, every smaller k's column comes from the k-prefixes of the matched k_max-mers. "
                        "Works with either --hash_mode.")
    p.add_argument('--sketch_hash', choices=['canonical', 'forward'], default='canonical',
                   help="with --reference_pipeline: what SELECTS a genome's n k-mers. canonical (default): the hash they match by "
                        "(--hash_mode). forward: MurmurHash3(k-mer as it stands in the genome) %% 9999999999971, the k-mer kept as it "
                        "stands - CMash's training without reverse complements as recollected (unverified).")
    a = p.parse_args(argv)
    if a.sketch_hash != 'canonical' and not a.reference_pipeline:
        p.error('--sketch_hash forward needs --reference_pipeline')
    if a.prefix_tables and a.hash_mode != 'cmash':
        p.error('--prefix_tables needs --hash_mode cmash')
    if a.prefix_tables and a.reference_pipeline:
        p.error('--prefix_tables and --reference_pipeline are two different definitions of the smaller-k columns')
    if os.path.isdir(a.genomes):
        paths = sorted(os.path.join(a.genomes, f) for f in os.listdir(a.genomes)
                       if '.fna' in f or f.endswith(('.fa', '.fa.gz', '.fasta', '.fasta.gz')))
    else:
        with open(a.genomes) as fh:
            paths = [ln.strip() for ln in fh if ln.strip()]
    ks = [int(x) for x in a.ks.split(',')]
    if a.hash_mode == 'cmash' or a.sketch_hash == 'forward':
        # (that mode's kernels exist for a list of k; found out here, not at the first launch after the genomes have been read.  The
        # reference pipeline hashes at its largest k only — the smaller k are prefixes)
        from . import _hip
        hashed = [max(ks)] if (a.reference_pipeline or a.prefix_tables) else ks
        missing = [k for k in hashed if k not in _hip.hash_mode1_ks()]
        if missing:
            p.error('--hash_mode cmash / --sketch_hash forward are built for k in {%s}; not for k = %s (--hash_mode canonical takes every k from 1 to 64)'
                    % (', '.join(str(k) for k in _hip.hash_mode1_ks()), ', '.join(str(k) for k in missing)))
    if a.reference_pipeline:
        if len(ks) > 4:
            p.error('--reference_pipeline takes at most four k')
        build_reference_pipeline(paths, a.out_dir, ks, a.num_hashes, hash_mode=1 if a.hash_mode == 'cmash' else 0, sketch_hash=a.sketch_hash)
        return
    build(paths, a.out_dir, ks, a.num_hashes, hash_mode=1 if a.hash_mode == 'cmash' else 0, prefix_tables=a.prefix_tables)


if __name__ == '__main__':
    main()
