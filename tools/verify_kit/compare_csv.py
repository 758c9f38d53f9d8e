"""Column-by-column comparison of two containment CSVs (CMash's StreamingQueryDNADatabase.py output and this package's
select_db CSV: first column the organism file name, then one containment column per k, ascending k).

    python tools/verify_kit/compare_csv.py REFERENCE.csv OURS.csv [--tol 1e-9]

Prints, per k, the organisms whose value differs by more than --tol and the largest difference; the organisms present in one file
only; and, for every kind of difference, which recollected rule of DESIGN.md §2 it would contradict.  Exit status 0: identical
within --tol (row order aside)."""
import argparse
import csv
import os
import sys

MEANING = """What a difference would mean (DESIGN.md §2, "parity unpinned"):
  * a row only in REFERENCE / only in OURS ........ the row filter (`-c 0 --sensitive`: organisms with containment > 0 at the SMALLEST k) is
                                                     recollected wrongly, or a smaller-k column differs around zero
  * the LAST column (k_max) differs ................ (a) which k-mers a genome's sketch holds: hash mode (`build_db --hash_mode cmash` =
                                                     min(hash(kmer), hash(revcomp)) mod 9999999999971 is CMash's CountEstimator as recollected;
                                                     `--sketch_hash forward` = no reverse complements at training) — rebuild the table with the
                                                     other setting and compare again; (b) the read side: KMC's canonical counting with -ci2 -cs3
                                                     (the identity oracle, mgo_refpipe_count_kmers) — compare `kmc_dump` of the intersection with
                                                     tools/verify_kit's matched k-mer list
  * only the SMALLER-k columns differ .............. the streaming query's prefix rule (a k-prefix of a matched k_max-mer or of its reverse
                                                     complement counts for every genome whose sketch holds that prefix) is recollected wrongly
  * values differ by a constant factor per column .. the denominator (distinct k-prefixes of the genome's sketched k_max-mers) is wrong"""


def load(path):
    with open(path) as fh:
        rows = list(csv.reader(fh))
    head, out = rows[0], {}
    for r in rows[1:]:
        if r:
            out[os.path.basename(r[0])] = [float(x) for x in r[1:]]
    return head, out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("reference")
    ap.add_argument("ours")
    ap.add_argument("--tol", type=float, default=1e-9)
    a = ap.parse_args(argv)
    hr, ref = load(a.reference)
    ho, ours = load(a.ours)
    bad = 0
    if len(hr) != len(ho):
        print("different numbers of columns: %r against %r" % (hr, ho))
        bad += 1
    for name in sorted(set(ref) - set(ours)):
        print("only in REFERENCE: %s %r" % (name, ref[name]))
        bad += 1
    for name in sorted(set(ours) - set(ref)):
        print("only in OURS: %s %r" % (name, ours[name]))
        bad += 1
    ncol = min(len(hr), len(ho)) - 1
    for c in range(ncol):
        diffs = [(abs(ref[n][c] - ours[n][c]), n) for n in ref if n in ours and abs(ref[n][c] - ours[n][c]) > a.tol]
        if diffs:
            worst = max(diffs)
            print("column %s: %d organisms differ, the largest by %.3g (%s: %r against %r)"
                  % (hr[c + 1] if c + 1 < len(hr) else c, len(diffs), worst[0], worst[1], ref[worst[1]][c], ours[worst[1]][c]))
            bad += len(diffs)
    if bad:
        print()
        print(MEANING)
        return 1
    print("identical within %g: %d organisms x %d columns" % (a.tol, len(ref), ncol))
    return 0


if __name__ == "__main__":
    sys.exit(main())
