#!/bin/bash
# rocprofv3 kernel statistics of the probe (which kernels, how long) -> gpurun_out/r06/kcount_trace_<tag>.csv
set -u
TAG=${1:-x}; shift
OUT=gpurun_out/r06/kcount_trace_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/raw" -o run -- python3 tools/kcount_probe.py --no_hash_path --reps 5 "$@" > "$OUT/log.txt" 2>&1
F=$(find "$OUT/raw" -name "*kernel_stats.csv" | head -1)
python3 - "$F" > "$OUT.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
print("kernel,calls,avg_us,total_ms,pct")
for r in rows[:14]:
    n = r["Name"]
    m = re.search(r"mg::\(anonymous namespace\)::(\w+(?:<[^>]*>)?)", n) or re.search(r"mg::(\w+(?:<[^>]*>)?)", n)
    print("%s,%s,%.1f,%.2f,%s" % ((m.group(1) if m else n[:50]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
rm -rf "$OUT"
cat "$OUT.csv"
