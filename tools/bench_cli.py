"""Wall-clock of the kept command line on synthetic reads (files on disk -> subset DB / CAMI profile), i.e. ingest
INCLUDED: file read, PCIe, on-device parsing / tokenising, the kernels, the host CAMI tail.

    python tools/bench_cli.py [nreads] [ngenomes] [ks]       e.g.  python tools/bench_cli.py 10000000 10000 21,31,51

bench.py imports measure() for its `with_ingest` field and hands it the workload it has in memory already (BASELINE
configs[2]: 10M reads, the 10k-genome multi-k table, 12.5M SAM lines), so the headline workload is the one measured with
ingest.  The files are written with numpy (fixed-width records: 10M reads in seconds, not minutes of Python loops)."""
import argparse
import gzip
import os
import shutil
import struct
import sys
import zlib
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ACC = "NZ_SYN%06d.1"


def _digits(idx, width):
    """(len(idx), width) uint8 ASCII decimal digits, zero padded."""
    idx = np.asarray(idx, dtype=np.int64)
    out = np.empty((len(idx), width), dtype=np.uint8)
    for p in range(width):
        out[:, width - 1 - p] = (idx // (10 ** p)) % 10 + 48
    return out


QUAL_BINS = np.frombuffer(b"F:,#", dtype=np.uint8)  # a binning sequencer's four quality values (Q37 / 25 / 11 / 2) ...
QUAL_CUM = np.array([0.86, 0.94, 0.985, 1.0])        # ... and how often each turns up


def write_fastq(path, rb, n, L=150, block=1_000_000, qualities="binned"):
    """@r<9 digits>\\n<seq>\\n+\\n<qual>\\n per read.  qualities: "binned" — every base one of four values drawn at random (what a
    `.fq.gz` of a binning sequencer compresses like: 1.75 x the size of the constant file's); "constant" — all 'I' (rounds 1-5)."""
    seqs = rb.reshape(n, L)
    rng = np.random.default_rng(20260)
    w = 2 + 9 + 1 + L + 3 + L + 1
    with open(path, "wb") as fh:
        for a in range(0, n, block):
            b = min(a + block, n)
            rec = np.empty((b - a, w), dtype=np.uint8)
            rec[:, 0], rec[:, 1] = ord("@"), ord("r")
            rec[:, 2:11] = _digits(np.arange(a, b), 9)
            rec[:, 11] = 10
            rec[:, 12:12 + L] = seqs[a:b]
            rec[:, 12 + L:15 + L] = np.frombuffer(b"\n+\n", dtype=np.uint8)
            if qualities == "binned":
                rec[:, 15 + L:15 + 2 * L] = QUAL_BINS[np.searchsorted(QUAL_CUM, rng.random((b - a, L), dtype=np.float32))]
            else:
                rec[:, 15 + L:15 + 2 * L] = ord("I")
            rec[:, 15 + 2 * L] = 10
            rec.tofile(fh)
    return n * w


def write_sam(path, rb, src, n, G, L=150, block=1_000_000):
    """One primary line per read (strand from the read's parity) + a secondary (SEQ '*') to the next genome for every
    fourth read: 1.25 lines per read, fixed-width fields, written four reads at a time."""
    assert n % 4 == 0 and block % 4 == 0
    seqs = rb.reshape(n, L)
    tail = b"\t1000\t60\t%dM\t*\t0\t0\t" % L
    p_len = {0: None, 16: None}

    def primary(idx, flag):
        m = len(idx)
        head = b"r" + b"0" * 9 + b"\t" + (b"%d" % flag) + b"\t" + (ACC % 0).encode() + tail
        w = len(head) + L + 1 + L + len(b"\tNM:i:1\n")
        rec = np.empty((m, w), dtype=np.uint8)
        rec[:, :len(head)] = np.frombuffer(head, dtype=np.uint8)
        rec[:, 1:10] = _digits(idx, 9)
        acc0 = 10 + 1 + len(b"%d" % flag) + 1
        rec[:, acc0 + 6: acc0 + 12] = _digits(src[idx], 6)
        rec[:, len(head): len(head) + L] = seqs[idx]
        rec[:, len(head) + L] = 9
        rec[:, len(head) + L + 1: len(head) + 2 * L + 1] = ord("I")
        rec[:, len(head) + 2 * L + 1:] = np.frombuffer(b"\tNM:i:1\n", dtype=np.uint8)
        p_len[flag] = w
        return rec

    def secondary(idx):
        line = b"r" + b"0" * 9 + b"\t256\t" + (ACC % 0).encode() + b"\t1000\t0\t%dM10S\t*\t0\t0\t*\t*\tNM:i:5\n" % (L - 10)
        rec = np.empty((len(idx), len(line)), dtype=np.uint8)
        rec[:] = np.frombuffer(line, dtype=np.uint8)
        rec[:, 1:10] = _digits(idx, 9)
        rec[:, 15 + 6: 15 + 12] = _digits((src[idx] + 1) % G, 6)
        return rec

    total = 0
    with open(path, "wb") as fh:
        for a in range(0, n, block):
            b = min(a + block, n)
            i0 = np.arange(a, b, 4)
            parts = [primary(i0, 0), secondary(i0), primary(i0 + 1, 16), primary(i0 + 2, 0), primary(i0 + 3, 16)]
            out = np.concatenate(parts, axis=1)
            out.tofile(fh)
            total += out.size
    return total, n + n // 4


def write_data_dir(data, gb, go, G, glen, threads=16):
    """organism_files/*.fna.gz (one accession per genome), db_info.txt; -> (organism file names, accessions, subset db_info text)."""
    os.makedirs(os.path.join(data, "organism_files"))
    rows = ["Accession\tLength\tTaxID\tLineage\tTaxID_Lineage\n", "Unmapped\t0\tUnmapped\t|||||||Unmapped\t|||||||Unmapped\n"]
    accs, names = [], []
    for g in range(G):
        taxid = "%d.1" % (1000 + g)
        names.append("taxid_%s_genomic.fna.gz" % taxid.replace(".", "_"))
        accs.append(ACC % g)
        rows.append("\t".join([accs[g], str(glen), taxid, "Bacteria|P|C|O|F|G|S%d|S%d str" % (g, g), "2|1|2|3|4|5|%d|%s" % (1000 + g, taxid)]) + "\n")

    def one(g):
        blob = (">%s\n" % accs[g]).encode() + gb[int(go[g]):int(go[g + 1])].tobytes() + b"\n"
        with open(os.path.join(data, "organism_files", names[g]), "wb") as fh:
            fh.write(gzip.compress(blob, 1))

    with ThreadPoolExecutor(threads) as ex:  # (zlib releases the GIL)
        list(ex.map(one, range(G)))
    with open(os.path.join(data, "db_info.txt"), "w") as fh:
        fh.write("".join(rows[:1] + rows[2:]))
    return names, accs, "".join(rows)


def parallel_gzip(text, level=6, piece=16 << 20, threads=32):
    """ONE gzip member, compressed in pieces by many threads the way pigz does it: every piece is raw deflate primed with the 32 KB in
    front of it and ends on a sync flush (the last one finishes the stream), so the concatenation is one deflate stream."""
    mv = memoryview(text)
    cuts = list(range(0, len(mv), piece)) or [0]

    def one(i):
        a = cuts[i]
        b = min(a + piece, len(mv))
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY, bytes(mv[max(a - 32768, 0):a])) if a else zlib.compressobj(level, zlib.DEFLATED, -15)
        out = co.compress(mv[a:b])
        out += co.flush(zlib.Z_FINISH if i == len(cuts) - 1 else zlib.Z_SYNC_FLUSH)
        return out, zlib.crc32(mv[a:b]), b - a
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(one, range(len(cuts))))
    crc = 0
    for _, c, n in parts:
        crc = _crc_combine(crc, c, n)
    return b"\x1f\x8b\x08\0\0\0\0\0\0\x03" + b"".join(p[0] for p in parts) + struct.pack("<II", crc, len(mv) & 0xFFFFFFFF)


def _gf2_times(mat, vec):
    s, i = 0, 0
    while vec:
        if vec & 1:
            s ^= mat[i]
        vec >>= 1
        i += 1
    return s


def _crc_combine(crc1, crc2, len2):
    if len2 == 0:
        return crc1
    odd = [0xEDB88320] + [1 << i for i in range(31)]
    even = [_gf2_times(odd, odd[i]) for i in range(32)]
    odd = [_gf2_times(even, even[i]) for i in range(32)]
    while True:
        even = [_gf2_times(odd, odd[i]) for i in range(32)]
        if len2 & 1:
            crc1 = _gf2_times(even, crc1)
        len2 >>= 1
        if not len2:
            break
        odd = [_gf2_times(even, even[i]) for i in range(32)]
        if len2 & 1:
            crc1 = _gf2_times(odd, crc1)
        len2 >>= 1
        if not len2:
            break
    return crc1 ^ crc2


def parallel_bgzf(text, level=6, threads=32, block=65280):
    mv = memoryview(text)

    def one(a):
        c = mv[a:a + block]
        co = zlib.compressobj(level, zlib.DEFLATED, -15)
        raw = co.compress(c) + co.flush()
        return b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(raw) + 8 - 1) + raw + struct.pack("<II", zlib.crc32(c), len(c))
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(one, range(0, len(mv), block)))
    return b"".join(parts) + b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0\x1b\0\x03\0\0\0\0\0\0\0\0\0"


def measure(n=1_000_000, G=200, ks=(21,), reps=2, verbose=False, workload=None, glen=50_000, sketch_n=1000,
            definition="reference_pipeline", hash_mode=0, qualities="binned"):
    """workload: dict(gb, go, rb, src, + dbh, dbo | ref_arrays) — genomes, reads, source genome of every read, and the sketch
    table (per-k genome-major arrays, or the reference pipeline's arrays) — as bench.py builds them; None: generated here.
    definition / hash_mode: of the table written to disk (formats.py version 2 or 3); select_main follows the table."""
    from metalign_amd import formats, map_and_profile, select_db, synth
    from metalign_amd._hip import Hip
    hip = Hip.get()
    td = tempfile.mkdtemp(prefix="mg_cli_")
    try:
        t_gen = time.perf_counter()
        ref_arrays = dbh = dbo = None
        if workload is None:
            gb, go = synth.make_genomes(G, glen)
            rb, ro, src = synth.make_reads(gb, go, n, npresent=max(40, G // 20))
        else:
            gb, go, rb, src = (workload[x] for x in ("gb", "go", "rb", "src"))
            definition, hash_mode = workload.get("definition", definition), workload.get("hash_mode", hash_mode)
            ref_arrays, dbh, dbo = workload.get("ref_arrays"), workload.get("dbh"), workload.get("dbo")
        prev_mode = hip.hash_mode
        hip.set_hash_mode(hash_mode)
        try:
            if definition == "reference_pipeline" and ref_arrays is None:
                h, khi, klo, o = hip.sketch_genomes_kmers(gb, go, ks[-1], sketch_n)
                t = hip.refdb_build(h, khi, klo, o, list(ks))
                ref_arrays = t.download(kmers=True)
                t.free()
            elif definition != "reference_pipeline" and dbh is None:
                tables = [hip.sketch_genomes(gb, go, k, sketch_n) for k in ks]
                dbh, dbo = [t[0] for t in tables], [t[1] for t in tables]
        finally:
            hip.set_hash_mode(prev_mode)
        n -= n % 4
        data = os.path.join(td, "data")
        names, accs, sub_text = write_data_dir(data, gb, go, G, glen)
        if definition == "reference_pipeline":
            f = hip.filter_build(ref_arrays["pair_hash"])
            formats.write_refpipe_table(os.path.join(data, "sketch_table"), names, sketch_n, ref_arrays, f.download(), hash_mode=hash_mode)
            f.free()
        else:
            filters = {}
            for k, h in zip(ks, dbh):
                f = hip.filter_build(h)
                filters[k] = f.download()
                f.free()
            formats.write_sketch_table(os.path.join(data, "sketch_table"), names, list(ks), sketch_n,
                                       {k: (h, o) for k, h, o in zip(ks, dbh, dbo)}, filters, hash_mode=hash_mode)
        sub = os.path.join(td, "subset_db_info.txt")
        with open(sub, "w") as fh:
            fh.write(sub_text)
        fq = os.path.join(td, "reads.fq")
        fq_bytes = write_fastq(fq, rb[: n * 150], n, qualities=qualities)
        sam = os.path.join(td, "aln.sam")
        sam_bytes, sam_lines = write_sam(sam, rb[: n * 150], np.asarray(src[:n], dtype=np.int64), n, G)
        t_gen = time.perf_counter() - t_gen
        best = None
        for rep in range(reps):
            tmpd = os.path.join(td, "tmp%d" % rep)
            args = argparse.Namespace(reads=fq, data=data, cmash_results="NONE", cutoff=0.01, db="AUTO", db_dir="AUTO", dbinfo_in="AUTO",
                                      dbinfo_out="AUTO", input_type="AUTO", keep_temp_files=True, strain_level=False, temp_dir=tmpd,
                                      threads=4, sketch_table="AUTO", min_count=2, sketch_size=0)
            hip.prof_reset()
            prof = None
            if os.environ.get("MG_CLI_PROFILE") == "1" and rep == reps - 1:  # cProfile of the last repetition's two stages
                import cProfile
                prof = cProfile.Profile()
                prof.enable()
            t0 = time.perf_counter()
            sk_t0 = time.perf_counter()
            select_db.run_timings = {}
            select_db.select_main(args)
            t1 = time.perf_counter()
            a2 = argparse.Namespace(infiles=[sam], data=data, db="NONE", dbinfo=sub, input_type="AUTO", length_normalize=False, low_mem=False,
                                    min_abundance=1e-4, rank_renormalize=False, output=os.path.join(td, "ab.tsv"), pct_id=0.5,
                                    no_quantify_unmapped=False, read_cutoff=1, sampleID="x", threads=4, verbose=False)
            map_and_profile.map_main(a2)
            t2 = time.perf_counter()
            if prof is not None:
                import io
                import pstats
                prof.disable()
                buf = io.StringIO()
                pstats.Stats(prof, stream=buf).sort_stats("cumulative").print_stats(40)
                print(buf.getvalue()[:8000], flush=True)
            tm = dict(getattr(select_db, "run_timings", {}))
            if verbose:
                print("run %d: select_main %.3f s %s, map_main %.3f s" % (rep, t1 - t0, {k: round(v, 3) for k, v in tm.items()}, t2 - t1), flush=True)
            if best is None or (t2 - t0) < best[0] + best[1]:
                best = (t1 - t0, t2 - t1, tm)
            del sk_t0
        sel, mp, tm = best
        nsel = sum(1 for _ in open(os.path.join(td, "tmp0", "subset_db_info.txt"))) - 2
        res = {"reads": n, "genomes": G, "ks": list(ks), "stage_a_definition": definition, "hash_mode": hash_mode, "fastq_mb": fq_bytes >> 20, "sam_mb": sam_bytes >> 20, "sam_lines": sam_lines,
               "select_main_s": sel, "map_main_s": mp, "generate_s": t_gen, "selected_genomes": nsel,
               "select_main_reads_per_s": n / sel, "map_main_reads_per_s": n / mp,
               "value": n / (sel + mp), "unit": "reads/s",
               "what": "metalign_amd.select_db.select_main (FASTQ file -> reader threads -> page-locked chunks -> HBM -> parse on "
                       "device -> ONE set of counting tables for all k, containment, CSV, cutoff, zcat of the selected genomes) + "
                       "map_and_profile.map_main (SAM file streamed the same way -> tokenise on device, assign, multimapped "
                       "resolution, CAMI file), files in the page cache, best of %d" % reps}
        # select_main on the reads as the reference usually gets them: `.fq.gz` (scripts/select_db.py:146-148), one gzip member, inflated
        # on the device (mg_inflate.hip).  (The replay's SAM file stays plain: the reference opens it as text.)
        try:
            t_z = time.perf_counter()
            nthreads = min(os.cpu_count() or 8, 64)
            with open(fq + ".gz", "wb") as fh:
                fh.write(parallel_gzip(np.fromfile(fq, dtype=np.uint8), threads=nthreads))
            t_z = time.perf_counter() - t_z
            gbest = None
            for rep in range(reps):
                tmpd = os.path.join(td, "tmpgz%d" % rep)
                args = argparse.Namespace(reads=fq + ".gz", data=data, cmash_results="NONE", cutoff=0.01, db="AUTO", db_dir="AUTO", dbinfo_in="AUTO",
                                          dbinfo_out="AUTO", input_type="AUTO", keep_temp_files=True, strain_level=False, temp_dir=tmpd,
                                          threads=4, sketch_table="AUTO", min_count=2, sketch_size=0)
                t0 = time.perf_counter()
                select_db.select_main(args)
                t1 = time.perf_counter()
                if gbest is None or (t1 - t0) < gbest:
                    gbest = t1 - t0
            same = open(os.path.join(td, "tmpgz0", "subset_db_info.txt"), "rb").read() == open(os.path.join(td, "tmp0", "subset_db_info.txt"), "rb").read()
            res["gz"] = {"fastq_gz_mb": os.path.getsize(fq + ".gz") >> 20, "qualities": qualities, "select_main_s": gbest, "select_main_reads_per_s": n / gbest,
                         "value": n / (gbest + mp), "unit": "reads/s", "selection_identical_to_plain_file": same, "compress_s": t_z,
                         "what": "select_main on reads.fq.gz (one gzip member, level 6: compressed bytes over PCIe, inflated on the device, "
                                 "mg_inflate.hip) + the plain run's map_main, files in the page cache, best of %d" % reps}
            # ... and the alignments as samtools / bgzip leave them: BGZF (every 64 KB block its own deflate stream: no window to carry)
            t_z = time.perf_counter()
            sam_z = os.path.join(td, "aln_bgzf.sam")  # (the command line tells SAM input by the name's ending; the stream code by the content)
            with open(sam_z, "wb") as fh:
                fh.write(parallel_bgzf(np.fromfile(sam, dtype=np.uint8), threads=nthreads))
            t_z = time.perf_counter() - t_z
            mbest = None
            for rep in range(reps):
                a2 = argparse.Namespace(infiles=[sam_z], data=data, db="NONE", dbinfo=sub, input_type="AUTO", length_normalize=False, low_mem=False,
                                        min_abundance=1e-4, rank_renormalize=False, output=os.path.join(td, "ab_gz.tsv"), pct_id=0.5,
                                        no_quantify_unmapped=False, read_cutoff=1, sampleID="x", threads=4, verbose=False)
                t0 = time.perf_counter()
                map_and_profile.map_main(a2)
                t1 = time.perf_counter()
                mbest = t1 - t0 if mbest is None or t1 - t0 < mbest else mbest
            same_profile = open(os.path.join(td, "ab_gz.tsv"), "rb").read() == open(os.path.join(td, "ab.tsv"), "rb").read()
            res["gz"].update({"sam_bgzf_mb": os.path.getsize(sam_z) >> 20, "map_main_bgzf_s": mbest, "profile_identical_to_plain_file": same_profile,
                              "sam_compress_s": t_z, "value_both_compressed": n / (gbest + mbest),
                              "what_bgzf": "map_main on the BGZF-compressed SAM file (level 6, inflated on the device); value_both_compressed = reads / "
                                           "(select_main on the .fq.gz + map_main on the .sam.gz)"})
        except BaseException as e:  # noqa: BLE001  (a secondary figure; sys.exit of the command line included)
            res["gz"] = {"error": repr(e)}
        if tm.get("stream_s"):
            res["stage_a_b_from_file"] = {"seconds": tm["stream_s"] + tm.get("containment_s", 0.0),
                                         "reads_per_s": n / (tm["stream_s"] + tm.get("containment_s", 0.0)),
                                         "fastq_GBs": fq_bytes / tm["stream_s"] / 1e9, "stream_s": tm["stream_s"],
                                         "containment_s": tm.get("containment_s"), "table_load_s": tm.get("table_load_s"),
                                         "host_tail_s": tm.get("host_tail_s"),
                                         "what": "reads file -> read sketches of every k (the pipeline of mg_stream.hip) + stage B per k: "
                                                 "the part of select_main between the sketch table's load and the CSV"}
        return res
    finally:
        shutil.rmtree(td, ignore_errors=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    G = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    ks = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (21,)
    import json
    print(json.dumps(measure(n, G=G, ks=ks, verbose=True, definition=os.environ.get("MG_DEFINITION", "reference_pipeline"),
                             hash_mode=int(os.environ.get("MG_HASH_MODE", "0"))), indent=1))
