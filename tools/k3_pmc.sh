#!/bin/bash
# SQ counters of k_profile_pass, direct bins against forced hashed bins on the same records (tools/k3_probe.py args).
# Runs on the GPU box: bash tools/k3_pmc.sh R G present
set -u
export TMPDIR=/tmp
OUT=gpurun_out/k3_pmc; rm -rf $OUT; mkdir -p $OUT
for mode in direct hashed; do
  if [ $mode = hashed ]; then export MG_DEBUG_K3_HASHED=1; else unset MG_DEBUG_K3_HASHED || true; fi
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf $OUT/${mode}_${tag}
    timeout -s KILL 200 rocprofv3 --output-format csv --pmc $set --kernel-trace -d $OUT/${mode}_${tag} -o run -- python3 tools/k3_probe.py "$@" > $OUT/$mode.$tag.log 2>&1
    f=$(find $OUT/${mode}_${tag} -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$mode" <<'PY'
import csv, sys, collections
f, mode = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    if 'k_profile_pass' in r['Kernel_Name'] and 'Lb1' in r['Kernel_Name'] or 'k_profile_pass<true>' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in sorted(acc.items()):
    print("%-7s %-24s per launch %.4g  (%d launches)" % (mode, k, v / n, n))
PY
    rm -rf $OUT/${mode}_${tag}
  done
done
