"""Wall-clock of the kept command line on 1M synthetic reads (file -> CAMI), i.e. ingest INCLUDED.
python tools/bench_cli.py [nreads]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metalign_amd import build_db, map_and_profile, select_db, synth  # noqa: E402
from metalign_amd._hip import Hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
G = 200
hip = Hip.get(0)
td = tempfile.mkdtemp()
gb, go = synth.make_genomes(G, 50_000)
rb, ro, src = synth.make_reads(gb, go, n, npresent=40)
seqs = rb.reshape(n, 150)
data = os.path.join(td, "data")
os.makedirs(os.path.join(data, "organism_files"))
rows = ["Accession\tLength\tTaxID\tLineage\tTaxID_Lineage\n", "Unmapped\t0\tUnmapped\t|||||||Unmapped\t|||||||Unmapped\n"]
accs, names = [], []
import gzip
for g in range(G):
    taxid = "%d.1" % (1000 + g)
    nm = "taxid_%s_genomic.fna.gz" % taxid.replace(".", "_")
    acc = "NZ_SYN%06d.1" % g
    with gzip.open(os.path.join(data, "organism_files", nm), "wb", compresslevel=1) as fh:
        fh.write((">%s\n" % acc).encode() + gb[int(go[g]):int(go[g + 1])].tobytes() + b"\n")
    rows.append("\t".join([acc, "50000", taxid, "Bacteria|P|C|O|F|G|S%d|S%d str" % (g, g), "2|1|2|3|4|5|%d|%s" % (1000 + g, taxid)]) + "\n")
    accs.append(acc); names.append(nm)
open(os.path.join(data, "db_info.txt"), "w").write("".join(rows[:1] + rows[2:]))
sub = os.path.join(td, "subset_db_info.txt")
open(sub, "w").write("".join(rows))
t0 = time.perf_counter()
build_db.build([os.path.join(data, "organism_files", x) for x in names], os.path.join(data, "sketch_table"), [21], 1000)
print("build_db (200 genomes x 50 kb, k=21): %.2f s" % (time.perf_counter() - t0))
fq = os.path.join(td, "reads.fq")
with open(fq, "wb") as fh:
    qual = b"\n+\n" + b"I" * 150 + b"\n"
    fh.write(b"".join(b"@r%d\n" % i + seqs[i].tobytes() + qual for i in range(n)))
sam = os.path.join(td, "aln.sam")
with open(sam, "w") as fh:
    for i in range(n):
        s = seqs[i].tobytes().decode()
        fh.write("r%d\t%d\t%s\t1000\t60\t150M\t*\t0\t0\t%s\t%s\tNM:i:1\n" % (i, 16 * (i & 1), accs[src[i]], s, "I" * 150))
        if i % 4 == 0:
            fh.write("r%d\t256\t%s\t1000\t0\t140M10S\t*\t0\t0\t*\t*\tNM:i:5\n" % (i, accs[(src[i] + 1) % G]))
for rep in range(2):
    tmpd = os.path.join(td, "tmp%d" % rep)
    args = argparse.Namespace(reads=fq, data=data, cmash_results="NONE", cutoff=0.01, db="AUTO", db_dir="AUTO", dbinfo_in="AUTO",
                              dbinfo_out="AUTO", input_type="AUTO", keep_temp_files=True, strain_level=False, temp_dir=tmpd,
                              threads=4, sketch_table="AUTO", min_count=2, sketch_size=0)
    t0 = time.perf_counter()
    select_db.select_main(args)
    t1 = time.perf_counter()
    a2 = argparse.Namespace(infiles=[sam], data=data, db="NONE", dbinfo=sub, input_type="AUTO", length_normalize=False, low_mem=False,
                            min_abundance=1e-4, rank_renormalize=False, output=os.path.join(td, "ab.tsv"), pct_id=0.5,
                            no_quantify_unmapped=False, read_cutoff=1, sampleID="x", threads=4, verbose=False)
    if rep == 1 and os.environ.get("MG_CPROFILE"):
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        map_and_profile.map_main(a2)
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
    else:
        map_and_profile.map_main(a2)
    t2 = time.perf_counter()
    print("run %d: select_db.select_main (FASTQ %d MB -> subset db) %.3f s = %.2e reads/s;  map_main (SAM %d MB -> CAMI) %.3f s = %.2e reads/s"
          % (rep, os.path.getsize(fq) >> 20, t1 - t0, n / (t1 - t0), os.path.getsize(sam) >> 20, t2 - t1, n / (t2 - t1)))
print(open(os.path.join(td, "ab.tsv")).read()[:400])
