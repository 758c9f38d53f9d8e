"""Deterministic synthetic db_info + SAM text for the stage-C parity tests.

Everything is derived from `random.Random(seed)`, so the build container (where
tests/golden/make_golden.py feeds the text to the REFERENCE) and the GPU box
(where the tests feed the same text to the HIP path) see identical bytes.  Only
the expected outputs are committed; the bulk SAM text is regenerated on demand.

SAM columns follow the subset map_and_process reads
(/root/reference/scripts/map_and_profile.py:87,97,142-144,211,217).
"""
import random

DBINFO_HEADER = "Accesion\tLength\tTaxID\tLineage\tTaxID_Lineage\n"
UNMAPPED_ROW = "Unmapped\t0\tUnmapped\t|||||||Unmapped\t|||||||Unmapped\n"


def make_dbinfo(seed=7, n_species=12, max_strains=3, max_contigs=3):
    """-> (text, accessions[list], taxid_of_accession[list]).

    Taxa: strain-level genomes (several strains per species), a few genomes
    annotated only to species level (trailing empty strain field -> the '.0'
    unknown-strain path of gen_lower_taxa :344-364), and a virus whose middle
    ranks are empty.
    """
    rng = random.Random(seed)
    rows, accs, taxids = [], [], []
    acc_no = 0

    def add(taxid, namelin, taxlin, ncontigs):
        nonlocal acc_no
        for _ in range(ncontigs):
            acc_no += 1
            acc = "NZ_SYN%06d.1" % acc_no
            length = rng.randrange(40000, 90000)
            rows.append("\t".join([acc, str(length), taxid, namelin, taxlin]) + "\n")
            accs.append(acc)
            taxids.append(taxid)

    for s in range(n_species):
        phylum, cls, order, family, genus = 100 + s % 3, 200 + s % 4, 300 + s % 5, 400 + s % 6, 500 + s // 2
        species = 1000 + s
        names = ["Bacteria", "Phy%d" % phylum, "Cls%d" % cls, "Ord%d" % order, "Fam%d" % family,
                 "Genus%d" % genus, "Genus%d species%d" % (genus, species)]
        ids = ["2", str(phylum), str(cls), str(order), str(family), str(genus), str(species)]
        if s % 5 == 4:  # species-level only
            add(str(species), "|".join(names + [""]), "|".join(ids + [""]), rng.randrange(1, max_contigs + 1))
            continue
        for st in range(rng.randrange(1, max_strains + 1)):
            strain = 900000 + s * 10 + st
            add(str(strain), "|".join(names + ["%s str. %d" % (names[-1], st)]),
                "|".join(ids + [str(strain)]), rng.randrange(1, max_contigs + 1))
    # a virus with unassigned middle ranks
    add("77001", "Viruses|||||Vgenus|Vgenus virus A|Vgenus virus A isolate 1", "10239|||||7700|77000|77001", 1)
    return DBINFO_HEADER + UNMAPPED_ROW + "".join(rows), accs, taxids


_BASES = "ACGT"


def _seq(rng, n):
    return "".join(rng.choice(_BASES) for _ in range(n))


def _line(qname, flag, rname, cigar, seq, nm):
    qual = "*" if seq == "*" else "I" * len(seq)
    return "\t".join([qname, str(flag), rname, "1000", "60", cigar, "*", "0", "0", seq, qual, "NM:i:%d" % nm]) + "\n"


def _cigar_ok(rng, n):
    style = rng.randrange(4)
    if style == 0:
        return "%dM" % n
    if style == 1:
        a = rng.randrange(1, n // 3)
        return "%dS%dM" % (a, n - a)
    if style == 2:
        a = rng.randrange(10, n - 10)
        return "%dM1I%dM" % (a, n - a - 1)
    a = rng.randrange(10, n - 10)
    return "%dM2D%dM" % (a, n - a)


def _cigar_bad(rng, n):
    m = rng.randrange(5, n // 3)  # matched fraction < 0.5 -> filtered at the default pct_id
    return "%dM%dS" % (m, n - m)


def make_sam_single(seed, nreads, accs, taxids, readlen=40, header=True):
    """Single-end stream: unique, secondary-bearing, filtered, unmapped, supplementary reads."""
    rng = random.Random(seed)
    by_tax = {}
    for a, t in zip(accs, taxids):
        by_tax.setdefault(t, []).append(a)
    taxa = sorted(by_tax)
    # skewed abundance: a handful of taxa dominate
    weights = [1.0 / (1 + i) ** 1.3 for i in range(len(taxa))]
    out = []
    if header:
        out.append("@HD\tVN:1.6\tSO:unsorted\n")
        for a in accs[:3]:
            out.append("@SQ\tSN:%s\tLN:50000\n" % a)
    for r in range(nreads):
        q = "read%d" % r
        t = rng.choices(taxa, weights)[0]
        acc = rng.choice(by_tax[t])
        strand = 16 if rng.random() < 0.5 else 0
        u = rng.random()
        seq = _seq(rng, readlen)
        if u < 0.04:  # unmapped
            out.append(_line(q, 4, "*", "*", seq, 0))
        elif u < 0.09:  # fails pct_id
            out.append(_line(q, strand, acc, _cigar_bad(rng, readlen), seq, 3))
        elif u < 0.62:  # unique
            out.append(_line(q, strand, acc, _cigar_ok(rng, readlen), seq, rng.randrange(4)))
        elif u < 0.92:  # primary + secondaries (SEQ '*')
            out.append(_line(q, strand, acc, _cigar_ok(rng, readlen), seq, rng.randrange(4)))
            for _ in range(rng.randrange(1, 4)):
                v = rng.random()
                t2 = t if v < 0.35 else rng.choices(taxa, weights)[0]
                cig = _cigar_bad(rng, readlen) if rng.random() < 0.15 else _cigar_ok(rng, readlen)
                out.append(_line(q, 256 | strand, rng.choice(by_tax[t2]), cig, "*", rng.randrange(6)))
        elif u < 0.96:  # primary + supplementary (chimeric)
            out.append(_line(q, strand, acc, _cigar_ok(rng, readlen), seq, 1))
            out.append(_line(q, 2048 | strand, rng.choice(accs), "%dM%dH" % (readlen // 2, readlen - readlen // 2),
                             seq[: readlen // 2], 0))
        else:  # secondary that is unmapped-looking / short line noise
            out.append(_line(q, strand, acc, _cigar_ok(rng, readlen), seq, 0))
            out.append(_line(q, 256, rng.choice(accs), "*", "*", 0))
    return "".join(out)


def make_sam_paired(seed, npairs, accs, taxids, readlen=36):
    """Paired-end stream as `minimap2 -ax sr` emits it: read-1 records, then read-2 records, same QNAME."""
    rng = random.Random(seed)
    by_tax = {}
    for a, t in zip(accs, taxids):
        by_tax.setdefault(t, []).append(a)
    taxa = sorted(by_tax)
    weights = [1.0 / (1 + i) ** 1.2 for i in range(len(taxa))]
    out = []
    for r in range(npairs):
        q = "pair%d" % r
        t = rng.choices(taxa, weights)[0]
        acc1, acc2 = rng.choice(by_tax[t]), rng.choice(by_tax[t])
        s1, s2 = _seq(rng, readlen), _seq(rng, readlen)
        u = rng.random()
        if u < 0.45:  # proper pair, same taxon (maybe different contigs)
            out.append(_line(q, 99, acc1, _cigar_ok(rng, readlen), s1, 1))
            out.append(_line(q, 147, acc2, _cigar_ok(rng, readlen), s2, 1))
        elif u < 0.55:  # ends disagree
            t2 = rng.choice(taxa)
            out.append(_line(q, 65, acc1, _cigar_ok(rng, readlen), s1, 1))
            out.append(_line(q, 129, rng.choice(by_tax[t2]), _cigar_ok(rng, readlen), s2, 2))
        elif u < 0.65:  # mate unmapped: 73 / 133
            out.append(_line(q, 73, acc1, _cigar_ok(rng, readlen), s1, 0))
            out.append(_line(q, 133, "*", "*", s2, 0))
        elif u < 0.72:  # read 1 unmapped, read 2 mapped: 69 / 137
            out.append(_line(q, 69, "*", "*", s1, 0))
            out.append(_line(q, 137, acc2, _cigar_ok(rng, readlen), s2, 0))
        elif u < 0.92:  # both ends with secondaries: 99 + 355..., 147 + 403...
            out.append(_line(q, 99, acc1, _cigar_ok(rng, readlen), s1, 1))
            alts = [rng.choices(taxa, weights)[0] for _ in range(rng.randrange(1, 3))]
            for t2 in alts:
                out.append(_line(q, 355, rng.choice(by_tax[t2]), _cigar_ok(rng, readlen), "*", 2))
            out.append(_line(q, 147, acc2, _cigar_ok(rng, readlen), s2, 1))
            for t2 in alts if rng.random() < 0.7 else [rng.choice(taxa)]:
                cig = _cigar_bad(rng, readlen) if rng.random() < 0.2 else _cigar_ok(rng, readlen)
                out.append(_line(q, 403, rng.choice(by_tax[t2]), cig, "*", 2))
        elif u < 0.96:  # one end filtered by pct_id
            out.append(_line(q, 99, acc1, _cigar_bad(rng, readlen), s1, 9))
            out.append(_line(q, 147, acc2, _cigar_ok(rng, readlen), s2, 1))
        else:  # both unmapped
            out.append(_line(q, 77, "*", "*", s1, 0))
            out.append(_line(q, 141, "*", "*", s2, 0))
    return "".join(out)
