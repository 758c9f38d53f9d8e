#!/bin/bash
# Stage C's commit pass under SQ counters (one --pmc pass, kernel trace only): is it issue-bound or waiting?
# Writes gpurun_out/k3_pmc/summary.txt: per kernel the counters averaged over its launches and the VALU issue share
#   SQ_INSTS_VALU x 4 cycles (a wave64 instruction on a 16-lane SIMD) / (GRBM_GUI_ACTIVE x 1024 SIMDs ... per XCD-summed counters).
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/k3_pmc"; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAVES"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -s KILL 600 rocprofv3 --output-format csv --pmc $set --kernel-trace -d "$OUT/$tag" -o run -- python3 tools/k3_probe.py 10000000 10000 500 > "$OUT/$tag.log" 2>&1
done
python3 - "$OUT" <<'PY' > "$OUT/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "k_profile_pass" not in k: continue
        k = "true (commit)" if "<true>" in k or "Lb1" in k else "false (map only)"
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k][row["Counter_Name"]] += 1
for k in acc:
    print("k_profile_pass<%s>: counters per launch (sum over XCDs / SEs as rocprofv3 reports them)" % k)
    v = {c: acc[k][c] / n[k][c] for c in acc[k]}
    for c in sorted(v): print("  %-24s %.4g   (%d launches)" % (c, v[c], n[k][c]))
    if "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v:
        simd_cycles = v["GRBM_GUI_ACTIVE"] / 8 * 1024  # GUI_ACTIVE summed over 8 XCDs; 256 CUs x 4 SIMDs
        print("  VALU issue share (4 cycles per wave64 instruction): %.3f" % (v["SQ_INSTS_VALU"] * 4 / simd_cycles))
PY
cat "$OUT/summary.txt"
