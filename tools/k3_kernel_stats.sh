#!/bin/bash
# Stage C's kernels alone under rocprofv3 --stats (tools/k3_probe.py's records: 12.5M over 10 001 taxa, hashed bins): average
# duration per kernel -> gpurun_out/k3stat/summary.txt
cd /tmp; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf "$R/gpurun_out/k3stat"
cd "$R" && MG_SINGLE_STREAM=1 rocprofv3 --output-format csv --kernel-trace --stats -d gpurun_out/k3stat -o run -- python3 tools/k3_probe.py 10000000 10000 500 > gpurun_out/k3stat_probe.txt 2>&1
grep -h "k_bins\|k_profile\|k_pass_prep" gpurun_out/k3stat/run_kernel_stats.csv | sed 's/(.*)",/",/' > gpurun_out/k3stat/summary.txt
cat gpurun_out/k3stat/summary.txt
