#!/bin/bash
# Stage A's kernel under SQ counters for the LDS side (two --pmc passes, kernel trace only): is k_sketch_reads held by its table
# look-ups?  -> gpurun_out/k1_lds_pmc/summary.txt
cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/k1_lds_pmc"; rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
export MG_SINGLE_STREAM=1
ARGS="--steps 4 --warmup 2 --no_cpu_baseline --no_secondary --no_kernel_table $*"
for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -s KILL 600 rocprofv3 --output-format csv --pmc $set --kernel-trace -d "$OUT/$tag" -o run -- python3 bench.py $ARGS > "$OUT/$tag.log" 2>&1
done
python3 - "$OUT" <<'PY' > "$OUT/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_sketch_reads" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
print("k_sketch_reads: counters per launch (sum over XCDs / SEs as rocprofv3 reports them)")
v = {c: acc[c] / n[c] for c in acc}
for c in sorted(v): print("  %-24s %.4g   (%d launches)" % (c, v[c], n[c]))
PY
cat "$OUT/summary.txt"
