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
