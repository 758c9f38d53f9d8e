"""Condenses the rocprofv3 outputs of tools/collect_profiles.sh into small files fit for profiles/<round>/:
kernel_stats.csv (library kernels only, names shortened), pmc_traffic.json (HBM bytes per launch per kernel,
corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are in KB; on gfx950
FETCH_SIZE counts half of a wide coalesced read) and pmc_sq.json (VALU issue statistics of the dominant kernel).
Usage: python tools/summarize_profiles.py gpurun_out/prof_<tag>"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"mg::(k_\w+(?:<[^>]*>)?)", name)
    if m:
        return m.group(1)
    if "rocprim" in name:
        return "rocprim::" + (re.search(r"detail::(\w+)", name).group(1) if re.search(r"detail::(\w+)", name) else "kernel")
    return name.split("(")[0][:60]


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def main(out):
    res = {}
    f = one(os.path.join(out, "stats", "**", "*kernel_stats.csv"))
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, "kernel_stats.csv"), "w") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        res["kernel_stats"] = {short(r["Name"]): float(r["AverageNs"]) for r in rows}
    traffic = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = one(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"))
        if not f:
            continue
        acc = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c:
                continue
            k = short(r["Kernel_Name"])
            a = acc.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        for k, (n, tot) in acc.items():
            kb = tot / n
            scale = 2048.0 if c == "FETCH_SIZE" else 1024.0  # KB -> bytes; FETCH_SIZE x2 on gfx950 (guide, HBM section)
            traffic.setdefault(k, {"launches": n})[c.lower() + "_bytes"] = kb * scale
    for k, d in traffic.items():
        d["hbm_bytes_per_launch"] = d.get("fetch_size_bytes", 0.0) + d.get("write_size_bytes", 0.0)
    if traffic:
        json.dump({"correction": "FETCH_SIZE KB x1024 x2 (gfx950 counts half of wide coalesced reads), WRITE_SIZE KB x1024; "
                                 "atomics are read-modify-writes at the memory side and show up in WRITE_SIZE",
                   "kernels": {k: v for k, v in traffic.items() if k.startswith("k_")}},
                  open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    f = one(os.path.join(out, "pmc_SQ", "**", "*counter_collection.csv"))
    if f:
        acc = {}
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            d = acc.setdefault(k, {})
            a = d.setdefault(r["Counter_Name"], [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        sq = {k: {c: tot / n for c, (n, tot) in d.items()} for k, d in acc.items()}
        for k, d in sq.items():
            if d.get("SQ_BUSY_CYCLES") and d.get("SQ_ACTIVE_INST_VALU"):
                d["valu_busy_frac_of_wave_cycles"] = d["SQ_ACTIVE_INST_VALU"] / max(d.get("SQ_WAVE_CYCLES", 1.0), 1.0)
        json.dump(sq, open(os.path.join(out, "pmc_sq.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in res.get("kernel_stats", {}).items() if k.startswith("k_")}, indent=1))
    print(json.dumps({k: v.get("hbm_bytes_per_launch") for k, v in traffic.items() if k.startswith("k_")}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
