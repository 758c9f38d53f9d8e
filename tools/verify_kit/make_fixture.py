"""The fixture of tools/verify_against_cmash.sh: twenty small genomes and a read set whose expected selection is known by
construction, written as the files both pipelines read.

    python tools/verify_kit/make_fixture.py OUT_DIR [--genomes 20] [--genome_len 60000] [--reads 40000] [--seed 7]

OUT_DIR/organism_files/taxid_<id>_genomic.fna.gz   one FASTA per genome (what setup_data.sh leaves)
OUT_DIR/training_files.txt                           their paths, one per line (MakeStreamingDNADatabase.py's input)
OUT_DIR/db_info.txt                                  the reference's db_info.txt layout (accession, length, taxid, lineage)
OUT_DIR/reads.fq                                     150 bp reads of SIX of the genomes (both strands, 1 % substitutions, a few N),
                                                     one genome at 0.5x coverage (below -ci2 almost everywhere: must NOT be selected)
OUT_DIR/expected.json                                what the design says the answer is: the present genomes, the absent ones, and
                                                     for every genome whether a containment of ~1, ~0 or "low" is expected at k = 60
Two of the genomes are 97 %-identical strains of each other (k = 30 columns must see both when one is present; k = 60 only one);
one genome is the reverse complement of another (both must score alike).  Pure numpy: no GPU, no reference code."""
import argparse
import gzip
import json
import os

import numpy as np

ALPHA = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = bytes.maketrans(b"ACGT", b"TGCA")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--genomes", type=int, default=20)
    ap.add_argument("--genome_len", type=int, default=60000)
    ap.add_argument("--reads", type=int, default=40000)
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args(argv)
    rng = np.random.default_rng(a.seed)
    G, L = a.genomes, a.genome_len
    assert G >= 10
    genomes = [ALPHA[rng.integers(0, 4, size=L)] for _ in range(G)]
    # genome 1 = a strain of genome 0 (3 % substitutions); genome 3 = the reverse complement of genome 2
    strain = genomes[0].copy()
    pos = rng.choice(L, size=int(0.03 * L), replace=False)
    strain[pos] = ALPHA[rng.integers(0, 4, size=len(pos))]
    genomes[1] = strain
    genomes[3] = np.frombuffer(bytes(genomes[2]).translate(COMP)[::-1], dtype=np.uint8).copy()
    os.makedirs(os.path.join(a.out, "organism_files"), exist_ok=True)
    names, lines = [], ["Accesion\tLength\tTaxID\tLineage\tTaxID_Lineage", "Unmapped\t0\tUnmapped\t|||||||Unmapped\t|||||||Unmapped"]
    for g, seq in enumerate(genomes):
        taxid, acc = 100000 + g, "NZ_VERIFY%04d.1" % g
        fn = os.path.join(a.out, "organism_files", "taxid_%d_genomic.fna.gz" % taxid)
        with gzip.open(fn, "wb", compresslevel=1) as fh:
            fh.write(b">" + acc.encode() + b" synthetic genome %d\n" % g)
            for i in range(0, L, 80):
                fh.write(bytes(seq[i:i + 80]) + b"\n")
        names.append(os.path.abspath(fn))
        sp = 0 if g < 2 else g  # (the strain pair is one species — select_db keeps one organism per species unless --strain_level)
        lineage = "k__Bacteria|p__P%d|c__C%d|o__O%d|f__F%d|g__G%d|s__S%d|t__T%d" % ((g // 4,) * 5 + (sp, g))
        ids = "|".join(str(x) for x in (2, 10 + g // 4, 20 + g // 4, 30 + g // 4, 40 + g // 4, 50 + g // 4, 1000 + sp, taxid))
        lines.append("%s\t%d\t%d\t%s\t%s" % (acc, L, taxid, lineage, ids))
    with open(os.path.join(a.out, "training_files.txt"), "w") as fh:
        fh.write("\n".join(names) + "\n")
    with open(os.path.join(a.out, "db_info.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    # reads: genomes 0, 2, 5, 6, 7 at ~15x each, genome 9 at 0.5x
    present, thin = [0, 2, 5, 6, 7], 9
    per = (a.reads - int(0.5 * L / 150)) // len(present)
    plan = [(g, per) for g in present] + [(thin, int(0.5 * L / 150))]
    with open(os.path.join(a.out, "reads.fq"), "wb") as fh:
        n = 0
        for g, cnt in plan:
            for _ in range(cnt):
                s = int(rng.integers(0, L - 150))
                r = bytearray(genomes[g][s:s + 150])
                if rng.random() < 0.5:
                    r = bytearray(bytes(r).translate(COMP)[::-1])
                for j in np.flatnonzero(rng.random(150) < 0.01):
                    r[j] = int(ALPHA[rng.integers(0, 4)])
                if rng.random() < 0.01:
                    r[int(rng.integers(0, 150))] = ord("N")
                fh.write(b"@v%07d\n" % n + bytes(r) + b"\n+\n" + b"F" * 150 + b"\n")
                n += 1
    expect = {"genomes": G, "reads": n, "present": present, "thin_coverage": [thin], "strain_of": {"1": 0}, "reverse_complement_of": {"3": 2},
              "expected_k60": {str(g): ("high" if g in present or g == 3 else ("low" if g in (1, thin) else "zero")) for g in range(G)},
              "note": "high: containment near the fraction of sketched 60-mers that survive 1 % read errors at this coverage (> 0.5); "
                      "genome 3 is genome 2's reverse complement: the same canonical 60-mers, the same score; genome 1 shares only the "
                      "60-mers its 3 % substitutions left intact (~0.16) but most 30-mers (the k = 30 column ~0.4); genome 9 is covered 0.5x: "
                      "almost none of its 60-mers occurs twice (-ci2); every other genome: 0"}
    with open(os.path.join(a.out, "expected.json"), "w") as fh:
        json.dump(expect, fh, indent=1)
    print("fixture written to %s: %d genomes, %d reads" % (a.out, G, n))


if __name__ == "__main__":
    main()
