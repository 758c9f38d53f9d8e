#!/bin/bash
# SQ counters of k_count_kmers on the probe's workload (own rocprofv3 passes: --pmc with --kernel-trace only).
# Usage: [KERNEL=k_match_items] bash tools/kcount_pmc.sh <tag> [probe args]   -> gpurun_out/r06/kcount_pmc_<tag>.txt
set -u
TAG=${1:-x}; shift
OUT=gpurun_out/r06/kcount_pmc_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
: > "$OUT.txt"
for SET in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT"; do
  rm -rf "$OUT/raw"
  rocprofv3 --output-format csv --pmc $SET --kernel-trace -d "$OUT/raw" -o run -- python3 tools/kcount_probe.py --no_hash_path --reps 2 "$@" > "$OUT/log.txt" 2>&1
  python3 - "$OUT" "${KERNEL:-k_count_kmers}" >> "$OUT.txt" <<'PY'
import csv, glob, sys
out, kern = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/raw/**/*counter_collection.csv", recursive=True)
acc = {}
for r in csv.DictReader(open(f[0])):
    if kern not in r["Kernel_Name"]:
        continue
    a = acc.setdefault(r["Counter_Name"], [0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"])
for c, v in acc.items():
    print("%-24s %16.0f  (mean of %d launches)" % (c, v[1] / v[0], v[0]))
PY
done
tail -2 "$OUT/log.txt" | cut -c1-400 >> "$OUT.txt"
rm -rf "$OUT/raw" "$OUT"
cat "$OUT.txt"
