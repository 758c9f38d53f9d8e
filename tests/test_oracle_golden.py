"""CPU suite: pins the ORACLE (and the package's host logic) to the reference's golden vectors."""
import json
import os

import numpy as np
import pytest

import oracle
import stage_c_checks as sc

GOLDEN = sc.GOLDEN


def _oracle_assign(recs, ref2tax, ntax, pct_id):
    return oracle.profile_assign(recs, ref2tax, ntax, pct_id)


def test_murmur3_known_answers():
    with open(os.path.join(GOLDEN, "murmur3_kat.json")) as fh:
        kat = json.load(fh)
    assert len(kat) >= 40
    for v in kat:
        assert oracle.murmur3_x64_128(bytes.fromhex(v["hex"]), v["seed"]) == (v["h1"], v["h2"]), v


def test_murmur3_published_vectors():
    # widely published values of MurmurHash3_x64_128 (e.g. mmh3.hash64 documentation), as unsigned
    assert oracle.murmur3_x64_128(b"", 0) == (0, 0)
    h1, h2 = oracle.murmur3_x64_128(b"foo", 0)
    assert (h1, h2) == ((-2129773440516405919) % 2**64, 9128664383759220103)


@pytest.mark.parametrize("name,idx,run", sc.hand_cases(), ids=lambda v: str(v) if not isinstance(v, dict) else "")
def test_stage_c_hand_cases(name, idx, run, monkeypatch, tmp_path):
    sc.check_hand_case(name, run, _oracle_assign, monkeypatch, tmp_path)


@pytest.mark.parametrize("name", ["single_3k", "paired_2k", "single_100k", "paired_40k"])
def test_stage_c_bulk(name, monkeypatch, tmp_path):
    spec = sc.load_bulk()[name]
    sam, dbp = sc.materialise_bulk(name, spec, tmp_path)
    for run in spec["runs"]:
        sc.check_run(sam, dbp, run, _oracle_assign, monkeypatch, tmp_path, mm_digest_only=True)


def test_oracle_ingest_matches_package_tokeniser(tmp_path):
    """Two independent SAM tokenisers (oracle / package) agree on the seeded streams."""
    import samgen
    from metalign_amd import map_and_profile as mp
    dbtext, accs, taxids = samgen.make_dbinfo(seed=3, n_species=9)
    acc_index = {"Unmapped": 0}
    acc_index.update({a: i + 1 for i, a in enumerate(accs)})
    for text in (samgen.make_sam_single(1, 800, accs, taxids), samgen.make_sam_paired(2, 500, accs, taxids)):
        a = oracle.sam_to_records(text.splitlines(True), acc_index)
        b = mp.tokenise_sam(text.splitlines(True), acc_index)
        assert a.dtype == b.dtype and np.array_equal(a, b)


def test_vectorised_multimap_resolution_equals_list_version(tmp_path):
    """resolve_multi_prop_csr (numpy, from the kernel's CSR) == preprocess_multimapped + resolve_multi_prop
    (the reference's list algorithm) on the seeded bulk streams, bit for bit when bases are integers."""
    import copy
    from metalign_amd import map_and_profile as mp
    spec = sc.load_bulk()["single_3k"]
    sam, dbp = sc.materialise_bulk("single_3k", spec, tmp_path)
    for ov in ({}, {"read_cutoff": 0}, {"length_normalize": True}, {"pct_id": 0.9, "read_cutoff": 10}):
        args = sc.make_args(sam, dbp, str(tmp_path / "o.tsv"), ov)
        a2i, t2i = mp.get_acc2info(args)
        with open(sam) as fh:
            t2a_l, mm_l, _ = mp.map_and_process(args, fh, a2i, t2i, _assign=_oracle_assign)
        with open(sam) as fh:
            t2a_c, mm_c, _ = mp.map_and_process(args, fh, a2i, t2i, _assign=_oracle_assign, _want_lists=False)
        assert [[k, v] for k, v in t2a_l.items()] == [[k, v] for k, v in t2a_c.items()]
        assert mp.multimapped_lists(mm_c, mm_c["taxids"]) == mm_l
        mm_l = mp.preprocess_multimapped(args, mm_l, t2a_l)
        t2a_l = {k: v for k, v in t2a_l.items() if v[0] > args.read_cutoff}
        t2a_c = {k: v for k, v in t2a_c.items() if v[0] > args.read_cutoff}
        want = mp.resolve_multi_prop(args, copy.deepcopy(t2a_l), mm_l, {}, t2i)
        got = mp.resolve_multi_prop_csr(args, copy.deepcopy(t2a_c), mm_c, t2i)
        assert list(want) == list(got)
        for k in want:
            if ov.get("length_normalize"):
                assert abs(want[k][1] - got[k][1]) <= 1e-12 * max(1.0, abs(want[k][1]))
            else:
                assert want[k] == got[k], k


def test_library_multimap_resolution_equals_numpy_bit_for_bit():
    """mg_multimapped_shares (the library's host routine behind resolve_multi_prop_csr) against the numpy version it
    replaced, on random CSRs: repeated taxa within a read, taxa without an entry (NaN), reads left empty, zero weights,
    fractional weights (length-normalised runs) and genome lengths — every taxon's sum bit for bit."""
    import argparse
    import copy
    from metalign_amd import map_and_profile as mp
    from metalign_amd import _hip as _hip_mod
    rng = np.random.default_rng(9)
    for trial in range(8):
        big = trial >= 6  # (a list long enough for the library's host THREADS: the same sums, every taxon in read order)
        T = int(rng.integers(3, 60)) if not big else 700
        taxids = ["t%d" % i for i in range(T)]
        nreads = int(rng.integers(0, 4000)) if not big else 40_000
        lens = rng.integers(0, 7, size=nreads)
        off = np.zeros(nreads + 1, dtype=np.uint64)
        off[1:] = np.cumsum(lens)
        tax = rng.integers(0, T, size=int(off[-1])).astype(np.uint32)
        hitlen = rng.integers(30, 151, size=nreads).astype(np.uint64)
        t2a, t2i = {}, {}
        for i, t in enumerate(taxids):
            t2i[t] = [float(rng.integers(1000, 9000)), "strain", "x|y"]
            if rng.random() < 0.7:
                w = 0.0 if rng.random() < 0.1 else (float(rng.integers(1, 10**6)) if trial % 2 == 0 else float(rng.random() * 1e3))
                t2a[t] = [int(rng.integers(1, 100)), w] + t2i[t]
        mm = dict(taxids=taxids, mm_offsets=off, mm_tax=tax, mm_hitlen=hitlen)
        for ln in (False, True):
            args = argparse.Namespace(verbose=False, length_normalize=ln)
            got = mp.resolve_multi_prop_csr(args, copy.deepcopy(t2a), mm, t2i)
            want = mp.resolve_multi_prop_csr_numpy(args, copy.deepcopy(t2a), mm, t2i)
            assert got == want, (trial, ln)
            for threads in (1, 2, 8):  # (forced: the short lists too, and the serial loop on the long ones)
                _hip_mod.debug_set("shares_threads", threads)
                try:
                    assert mp.resolve_multi_prop_csr(args, copy.deepcopy(t2a), mm, t2i) == want, (trial, ln, threads)
                finally:
                    _hip_mod.debug_set("shares_threads", 0)
