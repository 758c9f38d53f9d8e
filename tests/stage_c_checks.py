"""Shared assertions: the package's stage C against the golden vectors made by the reference.

`assign` is the record-level backend under test: the HIP library in the -m gpu tests, the CPU oracle in
the CPU tests (which then pin the ORACLE, and exercise the host logic, against the reference's outputs).
"""
import argparse
import hashlib
import json
import math
import os

import samgen
from metalign_amd import map_and_profile as mp

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_args(sam, dbinfo, out, overrides):
    a = argparse.Namespace(infiles=[sam], data="data/", db="NONE", dbinfo=dbinfo, input_type="sam",
                           length_normalize=False, low_mem=False, min_abundance=10 ** -4, rank_renormalize=False,
                           output=out, pct_id=0.5, no_quantify_unmapped=False, read_cutoff=1, sampleID="golden",
                           threads=4, verbose=False)
    for k, v in overrides.items():
        setattr(a, k, v)
    return a


def _same_value(got, want, approx):
    if isinstance(want, float) and approx:
        return math.isclose(got, want, rel_tol=1e-9, abs_tol=1e-12)
    return got == want and type(got) is type(want)


def _cami_close(got, want):
    """Line-by-line equality except the PERCENTAGE column, compared to 1e-6 (BASELINE.json tolerance)."""
    gl, wl = got.splitlines(), want.splitlines()
    assert len(gl) == len(wl), (len(gl), len(wl))
    for g, w in zip(gl, wl):
        gf, wf = g.split("\t"), w.split("\t")
        if len(wf) >= 5 and not w.startswith("@"):
            assert gf[:4] == wf[:4] and gf[5:] == wf[5:], (g, w)
            assert abs(float(gf[4]) - float(wf[4])) <= 1e-6 + 1e-5 + 1e-9, (g, w)  # %.5f quantisation + tolerance
        else:
            assert g == w


def check_run(sam_path, dbinfo_path, run, assign, monkeypatch, tmp_path, mm_digest_only=False):
    """One golden run (one flag set) of one SAM."""
    ov = run["args"]
    approx = bool(ov.get("length_normalize"))
    out = str(tmp_path / "abund.tsv")
    args = make_args(sam_path, dbinfo_path, out, ov)
    acc2info, taxid2info = mp.get_acc2info(args)
    raised = None
    try:
        with open(sam_path) as fh:
            t2a, mm, lowmem = mp.map_and_process(args, fh, acc2info, taxid2info, _assign=assign)
    except SystemExit as e:
        raised = ["SystemExit", str(e)]
    except Exception as e:  # noqa: BLE001
        raised = [type(e).__name__, str(e)]
    assert raised == run.get("map_and_process_raises"), (raised, run.get("map_and_process_raises"))
    if raised is None:
        got = [[k, v] for k, v in t2a.items()]
        want = run["taxids2abs"]
        assert [g[0] for g in got] == [w[0] for w in want], "taxon set / insertion order differs"
        for (k, gv), (_, wv) in zip(got, want):
            assert len(gv) == len(wv)
            for x, y in zip(gv, wv):
                assert _same_value(x, y, approx), (k, gv, wv)
        if mm_digest_only:
            assert len(mm) == run["multimapped_n"]
            assert hashlib.sha256(json.dumps(mm).encode()).hexdigest() == run["multimapped_sha256"]
        else:
            assert mm == run["multimapped"]
        assert lowmem == run["low_mem_mmap"]
    # whole stage through map_main -> CAMI text
    if assign is not None:
        monkeypatch.setattr(mp, "_device_assign", assign)
        monkeypatch.setattr(mp, "_device_tokenise", mp.tokenise_sam)
    args = make_args(sam_path, dbinfo_path, out, ov)
    raised = None
    try:
        mp.map_main(args)
    except SystemExit as e:
        raised = ["SystemExit", str(e)]
    except Exception as e:  # noqa: BLE001
        raised = [type(e).__name__, str(e)]
    assert raised == run.get("map_main_raises")
    if raised is None:
        with open(out) as fh:
            text = fh.read()
        if approx:
            _cami_close(text, run["cami"])
        else:
            assert text == run["cami"], "CAMI profile differs from the reference's"


def hand_cases():
    cdir = os.path.join(GOLDEN, "cases")
    names = sorted(f[:-5] for f in os.listdir(cdir) if f.endswith(".json"))
    out = []
    for n in names:
        with open(os.path.join(cdir, n + ".json")) as fh:
            runs = json.load(fh)
        for i, r in enumerate(runs):
            out.append((n, i, r))
    return out


def check_hand_case(name, run, assign, monkeypatch, tmp_path):
    cdir = os.path.join(GOLDEN, "cases")
    check_run(os.path.join(cdir, name + ".sam"), os.path.join(cdir, "dbinfo.txt"), run, assign, monkeypatch, tmp_path)


def load_bulk():
    with open(os.path.join(GOLDEN, "bulk.json")) as fh:
        return json.load(fh)


def materialise_bulk(name, spec, tmp_path):
    """Regenerate the seeded SAM + db_info text the golden run was made from (and check it is the same text)."""
    dbtext, accs, taxids = samgen.make_dbinfo(seed=spec["dbinfo_seed"], n_species=spec["dbinfo_species"])
    gen = samgen.make_sam_single if spec["kind"] == "single" else samgen.make_sam_paired
    text = gen(spec["seed"], spec["n"], accs, taxids)
    assert hashlib.sha256(text.encode()).hexdigest() == spec["sam_sha256"], "samgen drifted from the golden input"
    sam = tmp_path / (name + ".sam")
    sam.write_text(text)
    dbp = tmp_path / "dbinfo.txt"
    dbp.write_text(dbtext)
    return str(sam), str(dbp)
