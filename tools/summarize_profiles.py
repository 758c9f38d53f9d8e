"""Condenses the rocprofv3 outputs of tools/collect_profiles.sh into small files fit for profiles/<round>/:
kernel_stats.csv (library kernels only, names shortened), pmc_traffic.json (HBM bytes per launch and per PASS, corrected
as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts
half of a wide coalesced read) and pmc_sq_summary.json (VALU instruction counts per launch and per pass).  "Per pass" for
stage A = the sum over the k of one pass (one fused launch, or one launch per k).  bench.py reads the two JSON files of
the newest round whose workload matches its own.
Usage: python tools/summarize_profiles.py gpurun_out/prof_<tag>"""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")  # (mg_kcount.hip keeps its kernels in one)
    m = re.search(r"mg::(k_\w+(?:<.*>)?)", name)
    if m:
        return m.group(1).replace("mg::", "").replace(" ", "")
    if "rocprim" in name:
        m = re.search(r"detail::(\w+)", name)
        return "rocprim::" + (m.group(1) if m else "kernel")
    return name.split("(")[0][:60]


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def per_kernel(csvfile, counters):
    acc = {}
    for r in csv.DictReader(open(csvfile)):
        if r["Counter_Name"] not in counters:
            continue
        k = short(r["Kernel_Name"])
        a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: {c: {"launches": n, "avg": tot / n} for c, (n, tot) in d.items()} for k, d in acc.items()}


def stage_a_per_pass(d, key):
    """Sum over the stage-A kernels of one pass: k_count_kmers<K> (stage A by k-mer identity, the default since round 6) if the
    passes ran it; else the fused kernel if it ran, else one k_sketch_reads<K> per k."""
    kc = [k for k in d if k.startswith("k_count_kmers")]
    sk = [k for k in d if k.startswith("k_sketch_reads")]
    if kc and max(d[k].get("launches", 0) for k in kc) >= max([d[k].get("launches", 0) for k in sk] + [0]):
        top = max(kc, key=lambda k: d[k].get("launches", 0))
        return [top], d[top][key]
    fused = [k for k in d if k.startswith("k_sketch_reads_multi")]
    if len(fused) > 1:  # a job that measured index against filter at load ran both forms: the one its passes run
        fused = [max(fused, key=lambda k: d[k].get("launches", 0))]
    names = fused if fused else [k for k in d if re.match(r"k_sketch_reads<\d+", k)]
    if not fused and len(names) > 1:  # (the reference pipeline: ONE one-k launch per pass; a priming pass may have run another form)
        top = max(d[k].get("launches", 0) for k in names)
        names = [k for k in names if d[k].get("launches", 0) == top]
    return names, sum(d[k][key] for k in names)


def main(out):
    bench = {}
    for f in ("bench_under_rocprof.json", "bench.json"):
        p = os.path.join(out, f)
        if os.path.exists(p):
            lines = [ln for ln in open(p) if ln.startswith('{"metric"')]
            if lines:
                bench = json.loads(lines[-1])
                break
    wl = {}
    m = re.search(r"(\d+) synthetic 150bp reads/GPU vs (\d+)-genome .*k in \[([\d, ]+)\]", bench.get("config", {}).get("workload", ""))
    if m:
        wl = {"reads": int(m.group(1)), "genomes": int(m.group(2)), "ks": [int(x) for x in m.group(3).split(",")],
              "definition": bench.get("config", {}).get("stage_a_definition", "sketch_per_k"),
              "hash_mode": bench.get("config", {}).get("hash_mode", 0), "match": bench.get("config", {}).get("stage_a_match")}
    res = {}
    f = one(os.path.join(out, "stats", "**", "*kernel_stats.csv"))
    if f:
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(out, "kernel_stats.csv"), "w") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        res["kernel_stats_avg_ns"] = {short(r["Name"]): float(r["AverageNs"]) for r in rows if short(r["Name"]).startswith("k_")}
    f = one(os.path.join(out, "stats_pipelined", "**", "*kernel_stats.csv"))
    if f:  # the default multi-stream schedule: kernels of consecutive passes overlap, averages are not a kernel's own time
        with open(os.path.join(out, "pipelined_kernel_stats.csv"), "w") as fh:
            w = csv.writer(fh)
            w.writerow(["kernel", "calls", "total_ns", "avg_ns", "pct", "min_ns", "max_ns"])
            for r in csv.DictReader(open(f)):
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    traffic = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = one(os.path.join(out, "pmc_" + c, "**", "*counter_collection.csv"))
        if not f:
            continue
        # KB -> bytes.  On gfx950 FETCH_SIZE tallies a 128-byte request of a wide coalesced streaming read at 64 bytes (guide, HBM
        # section): such bytes are counted at HALF.  Other access widths are uncalibrated — the random 16-byte probes of a table are
        # NOT known to be under-counted, so doubling everything gives an upper bound, not a measurement: both are kept.
        for k, d in per_kernel(f, {c}).items():
            t = traffic.setdefault(k, {"launches": d[c]["launches"]})
            key = c.lower().replace("_size", "")
            t[key + "_bytes_raw"] = d[c]["avg"] * 1024.0
            t[key + "_bytes"] = d[c]["avg"] * (2048.0 if c == "FETCH_SIZE" else 1024.0)  # (x2 on every fetched byte: the upper bound)
    for k, d in traffic.items():
        d["hbm_bytes_per_launch"] = d.get("fetch_bytes", 0.0) + d.get("write_bytes", 0.0)
        d["hbm_bytes_per_launch_raw"] = d.get("fetch_bytes_raw", 0.0) + d.get("write_bytes", 0.0)
    if traffic:
        names, upper = stage_a_per_pass(traffic, "hbm_bytes_per_launch")
        _, raw = stage_a_per_pass(traffic, "hbm_bytes_per_launch_raw")
        # stage A's only wide coalesced streaming read is the stage copy of the reads (150 B of bases + 8 B of offsets per read, once
        # per pass): THAT share was counted at half; everything else (filter words, table slots: random 4-16 byte probes) as counted
        stream = 158.0 * wl.get("reads", 0)
        corrected = raw + min(stream / 2.0, sum(traffic[k].get("fetch_bytes_raw", 0.0) for k in names))
        doc = {"workload": wl,
               "correction": "hbm_bytes_per_pass = WRITE_SIZE KB x1024 + FETCH_SIZE KB x1024 with ONLY the streaming share (the stage copy "
                             "of the reads, 158 B/read, wide coalesced loads: counted at half on gfx950) doubled; _raw = as counted; _upper = "
                             "every fetched byte doubled (what rounds 1-4 reported); atomics are read-modify-writes at the memory side and "
                             "show up in WRITE_SIZE; two separate --pmc passes",
               "k_sketch_reads": {"kernels": names, "hbm_bytes_per_pass": corrected, "hbm_bytes_per_pass_raw": raw, "hbm_bytes_per_pass_upper": upper,
                                  "fetch_bytes_per_pass_raw": sum(traffic[k].get("fetch_bytes_raw", 0.0) for k in names),
                                  "fetch_bytes_per_pass_upper": sum(traffic[k].get("fetch_bytes", 0.0) for k in names),
                                  "write_bytes_per_pass": sum(traffic[k].get("write_bytes", 0.0) for k in names),
                                  "algorithmic_bytes_per_pass": 158 * wl.get("reads", 0)},
               "kernels": {k: v for k, v in traffic.items() if k.startswith("k_")}}
        doc["stage_a"] = doc["k_sketch_reads"]  # (the name bench.py reads since round 6; the older key stays for older readers)
        tot = corrected
        json.dump(doc, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
        res["stage_a_hbm_bytes_per_pass"] = tot
    f = one(os.path.join(out, "pmc_SQ", "**", "*counter_collection.csv"))
    if f:
        sq = {k: {c: v["avg"] for c, v in d.items()} | {"launches": max(v["launches"] for v in d.values())}
              for k, d in per_kernel(f, {"SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"}).items()
              if k.startswith("k_")}
        tr = one(os.path.join(out, "pmc_SQ", "**", "*kernel_trace.csv"))
        if tr:  # the launches' own durations in the SAME pass (for the engine clock: GRBM_GUI_ACTIVE / 8 / duration)
            acc = {}
            for r in csv.DictReader(open(tr)):
                a = acc.setdefault(short(r["Kernel_Name"]), [0, 0.0])
                a[0] += 1
                a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            for k in sq:
                if k in acc:
                    sq[k]["duration_ns"] = acc[k][1] / acc[k][0]
        names, tot = stage_a_per_pass(sq, "SQ_INSTS_VALU")
        doc = {"workload": wl, "k_sketch_reads": {"kernels": names, "SQ_INSTS_VALU_per_pass": tot,
                                                  "per_wave_step": tot / max(wl.get("reads", 1) * 150 / 64.0, 1.0)},
               "kernels": sq}
        doc["stage_a"] = doc["k_sketch_reads"]
        json.dump(doc, open(os.path.join(out, "pmc_sq_summary.json"), "w"), indent=1)
        res["stage_a_valu_insts_per_pass"] = tot
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
