#!/bin/bash
# SQ wait / LDS counters of the fused stage-A kernel on configs[2] (own rocprofv3 passes, --pmc with --kernel-trace only).
# Runs on the GPU box: bash tools/k1_pmc.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/k1_pmc; rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  tag=$(echo $set | cut -d' ' -f1)
  MG_SINGLE_STREAM=1 timeout -s KILL 300 rocprofv3 --output-format csv --pmc $set --kernel-trace -d $OUT/$tag -o run -- python3 bench.py --steps 4 --warmup 1 --no_cpu_baseline --no_secondary --no_kernel_table > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_sketch_reads_multi' in r['Kernel_Name']:
        a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
for k, (n, v) in sorted(acc.items()):
    print("%-26s per launch %.4g  (%d launches)" % (k, v / n, n))
PY
  rm -rf $OUT/$tag
done
