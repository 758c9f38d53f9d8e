#!/bin/bash
# SQ counters of a short bench run (own rocprofv3 pass, --pmc with --kernel-trace only).  Usage: bash tools/pmc_sq_quick.sh <tag> [bench args]
set -u
TAG=${1:-x}; shift
OUT=gpurun_out/sq_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/raw" -o run -- \
  python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_table "$@" > "$OUT/log.txt" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, re, json
out = sys.argv[1]
f = glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True)
acc = {}
for r in csv.DictReader(open(f[0])):
    m = re.search(r"mg::(k_\w+(?:<.*>)?)", r["Kernel_Name"])
    k = m.group(1) if m else r["Kernel_Name"][:40]
    d = acc.setdefault(k, {})
    a = d.setdefault(r["Counter_Name"], [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
res = {k: {c: v[1] / v[0] for c, v in d.items()} | {"launches": max(v[0] for v in d.values())} for k, d in acc.items() if k.startswith("k_")}
json.dump(res, open(out + "/pmc_sq.json", "w"), indent=1)
for k, d in sorted(res.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:8]:
    print(k[:70], {c: round(v) for c, v in d.items()})
PY
rm -rf "$OUT/raw"
