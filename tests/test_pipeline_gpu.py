"""-m gpu: the kept command line end to end on a synthetic data/ directory.

minimap2 is not on the path being accelerated and is absent from the image; a stub `minimap2` that replays
a canned SAM stands in for it so that the subprocess plumbing of map_and_profile (scripts/
map_and_profile.py:413-416) is exercised too."""
import gzip
import os
import stat

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def _make_data_dir(tmp_path, rng, ngen=12, glen=6000):
    data = tmp_path / "data"
    org = data / "organism_files"
    org.mkdir(parents=True)
    gb, go = util.random_genomes(rng, ngen, glen)
    rows = ["Accession\tLength\tTaxID\tLineage\tTaxID_Lineage\n"]
    names, accs = [], []
    for g in range(ngen):
        species = 1000 + g // 2  # two strains per species
        taxid = "%d.%d" % (species, g % 2 + 1)
        name = "taxid_%s_genomic.fna.gz" % taxid.replace(".", "_")
        acc = "NZ_G%03d.1" % g
        seq = bytes(gb[int(go[g]):int(go[g + 1])]).decode()
        half = len(seq) // 2
        with gzip.open(str(org / name), "wt") as fh:  # two contigs per genome
            fh.write(">%s c1\n%s\n>%s_b c2\n%s\n" % (acc, seq[:half], acc, seq[half:]))
        for a, ln in ((acc, half), (acc + "_b", len(seq) - half)):
            rows.append("\t".join([a, str(ln), taxid, "Bacteria|P|C|O|F|G%d|G%d s%d|G%d s%d str%d" % (g // 4, g // 4, species, g // 4, species, g % 2),
                                   "2|10|20|30|40|%d|%d|%s" % (500 + g // 4, species, taxid)]) + "\n")
        names.append(name)
        accs.append(acc)
    (data / "db_info.txt").write_text("".join(rows))
    return data, gb, go, names, accs


def test_build_db_select_and_profile(hip, oracle_lib, tmp_path, monkeypatch):
    from metalign_amd import build_db, formats, metalign
    rng = np.random.default_rng(2024)
    data, gb, go, names, accs = _make_data_dir(tmp_path, rng)
    ks, n = [21, 31], 150
    paths = [str(data / "organism_files" / nm) for nm in names]
    built = build_db.build(paths, str(data / "sketch_table"), ks, n)
    # the table equals the oracle's sketch of the same contigs joined by 'N'
    table = formats.SketchTable(str(data / "sketch_table"))
    assert table.names == names and table.ks == ks
    joined, offs = [], [0]
    for g in range(len(names)):
        s = gb[int(go[g]):int(go[g + 1])]
        half = len(s) // 2
        j = np.concatenate([s[:half], np.frombuffer(b"N", np.uint8), s[half:]])
        joined.append(j)
        offs.append(offs[-1] + len(j))
    jb, jo = np.concatenate(joined), np.asarray(offs, dtype=np.uint64)
    for k in ks:
        oh, oo = oracle_lib.sketch_genomes(jb, jo, k, n)
        h, o = table.arrays(k)
        assert np.array_equal(np.asarray(h), oh) and np.array_equal(o, oo)
        assert np.array_equal(built[k][0], oh)
    # reads from genomes 3 and 8 (strain 1000+1.2 / 1004.1), FASTQ
    rb, ro, src = util.sample_reads(rng, gb, go, 3000, 150, err=0.005, present=[3, 8])
    fq = tmp_path / "sample.fq"
    with open(fq, "w") as fh:
        for i in range(len(ro) - 1):
            s = bytes(rb[int(ro[i]):int(ro[i + 1])]).decode()
            fh.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    # canned SAM for the stub aligner: every read hits its source genome's first contig, 20 % also hit the other
    # present genome (a real aligner only sees the SUBSET database, so every RNAME is in the subset db_info)
    sam = tmp_path / "canned.sam"
    with open(sam, "w") as fh:
        fh.write("@HD\tVN:1.6\n")
        for i in range(len(ro) - 1):
            s = bytes(rb[int(ro[i]):int(ro[i + 1])]).decode()
            fh.write("\t".join(["r%d" % i, "0", accs[src[i]], "1", "60", "150M", "*", "0", "0", s, "I" * 150, "NM:i:0"]) + "\n")
            if i % 5 == 0:
                fh.write("\t".join(["r%d" % i, "256", accs[8 if src[i] == 3 else 3], "1", "0", "140M10S", "*", "0", "0", "*", "*", "NM:i:4"]) + "\n")
    stub = tmp_path / "bin"
    stub.mkdir()
    exe = stub / "minimap2"
    exe.write_text("#!/bin/sh\ncat %s\n" % sam)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(stub) + os.pathsep + os.environ["PATH"])
    out = tmp_path / "abundances.tsv"
    tmpd = tmp_path / "tmp"
    metalign.main([str(fq), str(data), "--output", str(out), "--temp_dir", str(tmpd), "--keep_temp_files",
                   "--sketch_table", str(data / "sketch_table"), "--sampleID", "s1"])
    # stage A/B: the CSV equals what the oracle computes for the same reads and table
    csv = (tmpd / "cmash_query_results.csv").read_text().splitlines()
    assert csv[0] == ",k=21,k=31"
    per_k = []
    for k in ks:
        h, o = table.arrays(k)
        qh, qc, tr, _ = oracle_lib.sketch_reads(rb, ro, k, hmax=int(np.asarray(h).max()))
        hits, sizes = oracle_lib.containment(qh, qc, tr, 2, np.asarray(h), o)
        per_k.append(hits / np.maximum(sizes, 1))
    got = {ln.split(",")[0]: [float(x) for x in ln.split(",")[1:]] for ln in csv[1:]}
    for g, nm in enumerate(names):
        if per_k[0][g] > 0:
            assert got[nm] == [float(per_k[0][g]), float(per_k[1][g])]
        else:
            assert nm not in got
    assert csv[1].split(",")[0] in (names[3], names[8])
    # selection: one strain per species above the 0.01 cutoff -> genomes 3 and 8 (their siblings share k-mers only by chance)
    sub = (tmpd / "subset_db_info.txt").read_text().splitlines()
    assert sub[0].startswith("Accesion") and sub[1].startswith("Unmapped")
    chosen = {ln.split("\t")[2] for ln in sub[2:]}
    assert {"1001.2", "1004.1"} <= chosen
    # stage C ran on the stub aligner's SAM against the SUBSET db_info
    text = out.read_text()
    assert text.startswith("@SampleID:s1\n@Version:Metalign\n")
    strains = [ln.split("\t") for ln in text.splitlines() if "\tstrain\t" in ln]
    assert {s[0] for s in strains} >= {"1001.2.1", "1004.1.1"}
    assert abs(sum(float(s[4]) for s in strains) - 100.0) < 1.0
    # the same table as a version-1 directory (genome-major files, no stored filter: inverted on the host when opened,
    # the filter built from the hashes): select_db writes the same CSV
    from metalign_amd import select_db
    v1 = tmp_path / "table_v1"
    formats.write_sketch_table_v1(str(v1), names, ks, n, {k: table.arrays(k) for k in ks})
    assert formats.SketchTable(str(v1)).version == 1 and table.version == 2
    tmp1 = tmp_path / "tmp_v1"
    args = select_db.select_parseargs([str(fq), str(data), "--temp_dir", str(tmp1), "--keep_temp_files", "--sketch_table", str(v1)])
    select_db.select_main(args)
    assert (tmp1 / "cmash_query_results.csv").read_text().splitlines() == csv
    # a reads file that does not fit the device goes through in record-aligned pieces whose sketches are merged
    # (MG_READ_BATCH_BYTES forces it here: ~40 pieces of the 1 MB file; a piece boundary falls inside records); .gz is
    # inflated on the host and cut the same way; FASTA pieces are cut at headers.  Same CSV every time.
    import gzip
    monkeypatch.setenv("MG_READ_BATCH_BYTES", "25000")
    monkeypatch.setenv("MG_PRIME_READS", "10")  # (and the table-sizing pass over a prefix of the first piece, which only large inputs take)
    gzq = tmp_path / "sample_gz.fq.gz"
    with open(fq, "rb") as src_fh, gzip.open(gzq, "wb") as dst:
        dst.write(src_fh.read())
    fa = tmp_path / "sample.fa"
    with open(fa, "w") as fh:
        for i in range(len(ro) - 1):
            seq = bytes(rb[int(ro[i]):int(ro[i + 1])]).decode()
            fh.write(">r%d\n%s\n%s\n" % (i, seq[:70], seq[70:]))  # sequences over two lines
    for j, reads_file in enumerate((fq, gzq, fa)):
        tmpj = tmp_path / ("tmp_batched%d" % j)
        args = select_db.select_parseargs([str(reads_file), str(data), "--temp_dir", str(tmpj), "--keep_temp_files",
                                           "--sketch_table", str(data / "sketch_table")])
        select_db.select_main(args)
        assert (tmpj / "cmash_query_results.csv").read_text().splitlines() == csv, reads_file
    monkeypatch.delenv("MG_READ_BATCH_BYTES")
    monkeypatch.delenv("MG_PRIME_READS")
    # the same command as ONE RANK of a torch.distributed.run launch (world size 1, every collective in the path:
    # MG_FORCE_DIST=1): reads taken by record-aligned byte range, table by hash range, distributed.ShardJob's exchange;
    # rank 0 writes the same CSV and the same subset db_info.  A child process: torch.distributed stays out of this one.
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for j, reads_file in enumerate((fq, fa)):
        tmpd2 = tmp_path / ("tmp_dist%d" % j)
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29551 + j), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                   MG_FORCE_DIST="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        r = subprocess.run([sys.executable, "-m", "metalign_amd.select_db", str(reads_file), str(data), "--temp_dir", str(tmpd2),
                            "--keep_temp_files", "--sketch_table", str(data / "sketch_table")],
                           capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert (tmpd2 / "cmash_query_results.csv").read_text().splitlines() == csv, reads_file
        assert (tmpd2 / "subset_db_info.txt").read_text().splitlines() == sub
    # ... and stage C the same way: the ranks tokenise the SAM text between them, rank 0 gathers the records and writes
    # the same CAMI file as the single process (which got these lines from the stub aligner)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29557", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MG_FORCE_DIST="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out2 = tmp_path / "abundances_dist.tsv"
    r = subprocess.run([sys.executable, "-m", "metalign_amd.map_and_profile", str(sam), str(data), "--dbinfo",
                        str(tmpd / "subset_db_info.txt"), "--output", str(out2), "--sampleID", "s1"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert out2.read_text() == text
    # ... and both command lines with two, three and eight ranks sharing this GPU (torch.distributed.run; gloo staged through
    # the host, since RCCL refuses several ranks on one device): the same CSV, subset db_info and CAMI file
    for world in (2, 3, 8):
        env = dict(os.environ, MG_DIST_BACKEND="gloo", MG_DIST_REPORT="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        launch = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                  "--master-addr", "127.0.0.1", "--master-port", str(29560 + world)]
        tmpw = tmp_path / ("tmp_world%d" % world)
        # (world 2 is handed the GZIPPED reads — rank 0 inflates them with the library's parallel inflater and scatters
        # record-aligned shares of the text, so EVERY rank sketches; rounds 2-3 had rank 0 sketch everything — world 3 the
        # FASTA file, world 8 the FASTQ file)
        r = subprocess.run(launch + ["-m", "metalign_amd.select_db", str({2: gzq, 3: fa}.get(world, fq)), str(data), "--temp_dir", str(tmpw), "--keep_temp_files",
                                     "--sketch_table", str(data / "sketch_table")],
                           capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert (tmpw / "cmash_query_results.csv").read_text().splitlines() == csv, world
        assert (tmpw / "subset_db_info.txt").read_text().splitlines() == sub
        shards = [int((tmpw / ("shard_rank%d.txt" % r)).read_text().split()[0]) for r in range(world)]
        assert sum(shards) == len(ro) - 1 and min(shards) > 0, (world, shards)  # every rank had reads of its own, .gz included
        outw = tmp_path / ("abundances_world%d.tsv" % world)
        r = subprocess.run(launch + ["-m", "metalign_amd.map_and_profile", str(sam), str(data), "--dbinfo",
                                     str(tmpd / "subset_db_info.txt"), "--output", str(outw), "--sampleID", "s1"],
                           capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert outw.read_text() == text, world


def test_exchange_path_on_one_gpu_under_rccl():
    """The multi-GPU choreography (all-gather, all-to-all, all-reduce over RCCL) with world size 1, plain steps and the
    software-pipelined run(), against the oracle.  A child process: torch.distributed state stays out of this one."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, os.path.join(here, "dist_single_rank.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "dist-single-rank ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    # the same with stage A's counting table forced to overflow on every pass: the words a rank publishes from its
    # pending sketch are then stale, every rank sees the flag and the all-gather is repeated after the rebuild
    env["MG_TEST_KNOBS"] = "distinct_hint_ppm=500"
    r = subprocess.run([sys.executable, os.path.join(here, "dist_single_rank.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0 and "dist-single-rank ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("world,overflow", [(2, False), (3, False), (8, False), (2, True)])
def test_several_ranks_on_one_gpu_with_the_real_kernels(world, overflow):
    """The multi-GPU path at world size > 1 with every kernel real: `world` processes share this GPU (stage A per read
    shard, slices by hash range, the merge of slices that come from different ranks, stage B on table slices, stage C
    with the carried state across shard edges; plain steps and four passes in flight; one k and the fused multi-k
    launch) and must give the oracle's unsharded results on every rank.  The collectives go over gloo with the tensors
    staged through the host (distributed._HostStagedGloo): RCCL refuses two ranks on one device, so its transport is the
    one thing this does not exercise (tests/dist_single_rank.py runs it at world size 1)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ)
    if overflow:  # every counting table undersized: sketches redone on the list path, the words all-gather repeated
        env["MG_TEST_KNOBS"] = "distinct_hint_ppm=500"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29590 + world + (5 if overflow else 0)),
                        os.path.join(here, "dist_two_ranks_one_gpu.py")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-6000:]
    for rank in range(world):
        assert "two-ranks-one-gpu ok (rank %d)" % rank in r.stdout


def test_bench_command_of_the_driver_at_two_ranks_on_one_gpu():
    """`python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`, the driver's multi-GPU command, with
    both ranks on this GPU (MG_DIST_BACKEND=gloo): rank 0 prints exactly ONE JSON line on stdout, whole-job value."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MG_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29597", os.path.join(root, "bench.py"), "--gpus", "2", "--config", "1",
                        "--reads", "200000", "--genomes", "200", "--steps", "5", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 200000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]



def test_select_main_on_a_cmash_mode_table_with_prefix_columns(hip, oracle_lib, tmp_path):
    """build_db --hash_mode cmash --prefix_tables, then select_main on it: the table records its mode, the reads are hashed by
    the same definition, and the CSV equals what the oracle computes under that definition (mode-1 sketch at k_max, k-prefix
    tables below it; DESIGN.md §2 — CMash as recollected, unverified)."""
    from metalign_amd import build_db, formats, select_db
    rng = np.random.default_rng(99)
    data, gb, go, names, accs = _make_data_dir(tmp_path, rng)
    ks, n = [21, 31, 40], 120  # (hash mode 1 is built for a list of k: mg_sketch_cmash.hip)
    paths = [str(data / "organism_files" / nm) for nm in names]
    build_db.build(paths, str(data / "sketch_table"), ks, n, hash_mode=1, prefix_tables=True)
    assert hip.hash_mode == 0  # (the builder puts the library's mode back)
    table = formats.SketchTable(str(data / "sketch_table"))
    assert table.hash_mode == 1 and table.prefix_tables
    # the same contigs as the builder sketches them: joined by 'N'
    joined, offs = [], [0]
    for g in range(len(names)):
        s = gb[int(go[g]):int(go[g + 1])]
        half = len(s) // 2
        j = np.concatenate([s[:half], np.frombuffer(b"N", np.uint8), s[half:]])
        joined.append(j)
        offs.append(offs[-1] + len(j))
    jb, jo = np.concatenate(joined), np.asarray(offs, dtype=np.uint64)
    oracle_lib.set_hash_mode(1)
    try:
        want_tabs = [oracle_lib.sketch_genomes_prefix(jb, jo, ks[-1], k, n) if k < ks[-1] else oracle_lib.sketch_genomes(jb, jo, k, n) for k in ks]
        for k, (oh, oo) in zip(ks, want_tabs):
            h, o = table.arrays(k)
            assert np.array_equal(np.asarray(h), oh) and np.array_equal(o, oo), k
        rb, ro, src = util.sample_reads(rng, gb, go, 3000, 150, err=0.005, present=[3, 8])
        fq = tmp_path / "sample.fq"
        with open(fq, "w") as fh:
            for i in range(len(ro) - 1):
                s = bytes(rb[int(ro[i]):int(ro[i + 1])]).decode()
                fh.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
        tmpd = tmp_path / "tmp"
        args = select_db.select_parseargs([str(fq), str(data), "--temp_dir", str(tmpd), "--keep_temp_files", "--sketch_table",
                                           str(data / "sketch_table")])
        select_db.select_main(args)
        csv = (tmpd / "cmash_query_results.csv").read_text().splitlines()
        assert csv[0] == ",k=21,k=31,k=40"
        per_k = []
        for k, (oh, oo) in zip(ks, want_tabs):
            qh, qc, tr, _ = oracle_lib.sketch_reads_filtered(rb, ro, k, oh, hmax=int(oh.max()))
            hits, sizes = oracle_lib.containment(qh, qc, tr, 2, oh, oo)
            per_k.append(hits / np.maximum(sizes, 1))
    finally:
        oracle_lib.set_hash_mode(0)
        hip.set_hash_mode(0)
    got = {ln.split(",")[0]: [float(x) for x in ln.split(",")[1:]] for ln in csv[1:]}
    for g, nm in enumerate(names):
        if per_k[0][g] > 0:
            assert got[nm] == [float(c[g]) for c in per_k], nm
        else:
            assert nm not in got
    assert csv[1].split(",")[0] in (names[3], names[8])



def _joined(gb, go, n):
    """The contigs of _make_data_dir's genomes as build_db sketches them: joined by 'N'."""
    joined, offs = [], [0]
    for g in range(n):
        s = gb[int(go[g]):int(go[g + 1])]
        half = len(s) // 2
        j = np.concatenate([s[:half], np.frombuffer(b"N", np.uint8), s[half:]])
        joined.append(j)
        offs.append(offs[-1] + len(j))
    return np.concatenate(joined), np.asarray(offs, dtype=np.uint64)


@pytest.mark.parametrize("mode,sketch_hash", [(0, "canonical"), (1, "canonical"), (0, "forward")],
                         ids=["canonical_kmer_hash", "cmash_recollection", "entries_selected_by_the_forward_hash"])
def test_select_main_on_a_reference_pipeline_table(hip, oracle_lib, tmp_path, mode, sketch_hash):
    """build_db --reference_pipeline, then select_main on it, as one process and as 1 / 2 / 3 ranks of a torch.distributed.run
    launch: the reads are sketched at the largest k only (what the reference's kmc call counts, scripts/select_db.py:50-52) and
    the CSV — same layout, one column per k — equals the oracle's reference pipeline; the selection and the subset database
    come out of the reference's own cutoff code unchanged."""
    import subprocess
    import sys
    from metalign_amd import build_db, formats, select_db
    rng = np.random.default_rng(5150 + mode)
    data, gb, go, names, accs = _make_data_dir(tmp_path, rng)
    ks, n = [21, 31, 41, 51], 150
    tdir = str(data / "sketch_table")
    build_db.main([str(data / "organism_files"), tdir, "-n", str(n), "-k", "21,31,41,51", "--reference_pipeline"] +
                  (["--hash_mode", "cmash"] if mode else []) + (["--sketch_hash", "forward"] if sketch_hash == "forward" else []))
    assert hip.hash_mode == 0
    table = formats.SketchTable(tdir)
    assert table.refpipe and table.hash_mode == mode and table.ks == ks and table.names == sorted(names) and table.sketch_hash == sketch_hash
    order = [names.index(nm) for nm in table.names]  # (build_db lists the directory sorted)
    jb, jo = _joined(gb, go, len(names))
    oracle_lib.set_hash_mode(mode)
    try:
        h, khi, klo, o = oracle_lib.sketch_genomes_kmers(jb, jo, ks[-1], n, sketch_hash=sketch_hash)
        # genomes in the table's order
        parts = [(h[int(o[g]):int(o[g + 1])], khi[int(o[g]):int(o[g + 1])], klo[int(o[g]):int(o[g + 1])]) for g in order]
        oo = np.zeros(len(order) + 1, np.uint64)
        oo[1:] = np.cumsum([len(p[0]) for p in parts])
        want = oracle_lib.refpipe_build(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
                                        np.concatenate([p[2] for p in parts]), oo, ks)
        got = table.refpipe_arrays()
        assert np.array_equal(got["pair_hash"], want["pair_hash"]) and np.array_equal(got["pair_gen"], want["pair_gen"])
        for t, k in zip(got["small"], ks[:-1]):
            for key in ("pa", "pb", "cid", "cgen", "gsize"):
                assert np.array_equal(np.asarray(t[key]), want["small"][k][key]), (k, key)
            assert t["nprefix"] == want["small"][k]["nprefix"]
        rb, ro, src = util.sample_reads(rng, gb, go, 4000, 150, err=0.005, present=[3, 8, 9])
        fq = tmp_path / "sample.fq"
        with open(fq, "w") as fh:
            for i in range(len(ro) - 1):
                s = bytes(rb[int(ro[i]):int(ro[i + 1])]).decode()
                fh.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
        qh, qc, _, _ = oracle_lib.sketch_reads(rb, ro, ks[-1], hmax=int(want["pair_hash"][-1]))
        hits, sizes = oracle_lib.refpipe_containment(qh, qc, 2, want)
    finally:
        oracle_lib.set_hash_mode(0)
    per_k = [hits[ki] / np.maximum(sizes[ki], 1) for ki in range(len(ks))]
    tmpd = tmp_path / "tmp"
    args = select_db.select_parseargs([str(fq), str(data), "--temp_dir", str(tmpd), "--keep_temp_files", "--sketch_table", tdir])
    select_db.select_main(args)
    assert hip.hash_mode == 0
    csv = (tmpd / "cmash_query_results.csv").read_text().splitlines()
    assert csv[0] == ",k=21,k=31,k=41,k=51"
    got = {ln.split(",")[0]: [float(x) for x in ln.split(",")[1:]] for ln in csv[1:]}
    for g, nm in enumerate(table.names):
        if per_k[0][g] > 0:
            assert got[nm] == [float(c[g]) for c in per_k], nm
        else:
            assert nm not in got
    top = {csv[i].split(",")[0] for i in (1, 2, 3)}
    assert top == {names[3], names[8], names[9]}
    sub = (tmpd / "subset_db_info.txt").read_text().splitlines()
    assert sub[0].startswith("Accesion") and len(sub) > 2
    # the same command as ranks of a torch.distributed.run launch: world 1 under RCCL (every collective in the path), 2 and 3
    # ranks sharing this GPU over gloo — the pairs by hash range, the count lists by prefix range, the bitmaps OR-ed
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for world in (1, 2, 3):
        tmpw = tmp_path / ("tmp_world%d" % world)
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        if world == 1:
            env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29620 + mode), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MG_FORCE_DIST="1")
            cmd = [sys.executable]
        else:
            env.update(MG_DIST_BACKEND="gloo")
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                   "127.0.0.1", "--master-port", str(29622 + 2 * world + mode)]
        r = subprocess.run(cmd + ["-m", "metalign_amd.select_db", str(fq), str(data), "--temp_dir", str(tmpw), "--keep_temp_files",
                                  "--sketch_table", tdir], capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert (tmpw / "cmash_query_results.csv").read_text().splitlines() == csv, world
        assert (tmpw / "subset_db_info.txt").read_text().splitlines() == sub


def test_select_main_takes_tables_of_five_k_and_of_descending_k(hip, oracle_lib, tmp_path):
    """A table of more than four k, and one whose k are not ascending (build_db neither sorts nor limits -k): the streamed
    sketch session takes 1..4 ascending k, so these go through the per-k launches — as one process and as a ShardJob (whose
    stage B takes its k four at a time)."""
    from metalign_amd import build_db, select_db
    from metalign_amd.distributed import ShardJob
    rng = np.random.default_rng(77)
    data, gb, go, names, accs = _make_data_dir(tmp_path, rng)
    jb, jo = _joined(gb, go, len(names))
    rb, ro, src = util.sample_reads(rng, gb, go, 3000, 150, err=0.005, present=[2, 7])
    fq = tmp_path / "sample.fq"
    with open(fq, "w") as fh:
        for i in range(len(ro) - 1):
            s = bytes(rb[int(ro[i]):int(ro[i + 1])]).decode()
            fh.write("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)))
    paths = [str(data / "organism_files" / nm) for nm in names]
    for ks in ([21, 31, 41, 51, 61], [60, 50, 40, 30]):
        tdir = str(tmp_path / ("table_%d" % len(ks)))
        build_db.build(paths, tdir, ks, 120)
        tmpd = tmp_path / ("tmp_%d" % len(ks))
        args = select_db.select_parseargs([str(fq), str(data), "--temp_dir", str(tmpd), "--keep_temp_files", "--sketch_table", tdir])
        select_db.select_main(args)
        csv = (tmpd / "cmash_query_results.csv").read_text().splitlines()
        assert csv[0] == "," + ",".join("k=%d" % k for k in ks)
        per_k, tabs = [], []
        for k in ks:
            oh, oo = oracle_lib.sketch_genomes(jb, jo, k, 120)
            tabs.append((oh, oo))
            qh, qc, tr, _ = oracle_lib.sketch_reads_filtered(rb, ro, k, oh, hmax=int(oh.max()))
            hits, sizes = oracle_lib.containment(qh, qc, tr, 2, oh, oo)
            per_k.append((hits, sizes))
        got = {ln.split(",")[0]: [float(x) for x in ln.split(",")[1:]] for ln in csv[1:]}
        for g, nm in enumerate(names):
            if per_k[0][0][g] > 0:
                assert got[nm] == [float(h[g] / max(z[g], 1)) for h, z in per_k], nm
        if ks == sorted(ks):  # (a job wants its k ascending)
            job = ShardJob(hip, None, 0, 1, k=ks)
            job.load(rb, ro, np.zeros(0, dtype=oracle_lib.REC_DTYPE), np.zeros(1, np.uint32), [t[0] for t in tabs], [t[1] for t in tabs], ntax=1)
            for out in (job.step(), job.run(3)):
                for ki in range(len(ks)):
                    assert np.array_equal(out["hits_k"][ki], per_k[ki][0]) and np.array_equal(out["sizes_k"][ki], per_k[ki][1]), ks[ki]


def test_this_package_s_side_of_the_verification_kit_gives_what_the_fixture_was_built_to_give(tmp_path):
    """tools/verify_kit/make_fixture.py, then the two command lines tools/verify_against_cmash.sh runs for this package (build_db
    --reference_pipeline --hash_mode cmash, select_db): the CSV's k = 60 column is high for the present genomes and the reverse-complemented
    copy, low for the 97 % strain and the 0.5x genome, absent or zero for the rest; the k = 30 column sees the strain."""
    import csv
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    kit = tmp_path / "kit"
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.run([sys.executable, os.path.join(root, "tools", "verify_kit", "make_fixture.py"), str(kit)], check=True, timeout=300)
    r = subprocess.run([sys.executable, "-m", "metalign_amd.build_db", str(kit / "training_files.txt"), str(kit / "ours" / "sketch_table"), "-n", "1000",
                        "-k", "30,40,50,60", "--reference_pipeline", "--hash_mode", "cmash"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    r = subprocess.run([sys.executable, "-m", "metalign_amd.select_db", str(kit / "reads.fq"), str(kit) + "/", "--sketch_table", str(kit / "ours" / "sketch_table"),
                        "--temp_dir", str(kit / "ours" / "tmp"), "--keep_temp_files", "--dbinfo_out", str(kit / "ours" / "subset_db_info.txt"),
                        "--db", str(kit / "ours" / "subset.fna")], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    expect = json.load(open(kit / "expected.json"))
    rows = list(csv.reader(open(kit / "ours" / "tmp" / "cmash_query_results.csv")))
    got = {int(os.path.basename(x[0]).split("_")[1]) - 100000: [float(v) for v in x[1:]] for x in rows[1:]}
    assert len(rows[0]) == 5  # name + four k
    for g in range(expect["genomes"]):
        kind, row = expect["expected_k60"][str(g)], got.get(g)
        if kind == "high":
            assert row is not None and row[-1] > 0.5, (g, row)
        elif kind == "zero":
            assert row is None or row[-1] == 0.0, (g, row)
        else:
            assert row is None or row[-1] < 0.35, (g, row)
    assert abs(got[2][-1] - got[3][-1]) < 0.05          # a genome and its reverse complement: the same canonical k-mers
    assert got[1][0] > got[1][-1] and got[1][0] > 0.2    # the strain: most 30-mers survive 3 % substitutions, few 60-mers do
    sel = open(kit / "ours" / "subset_db_info.txt").read()
    for g in expect["present"]:
        assert "NZ_VERIFY%04d.1" % g in sel
