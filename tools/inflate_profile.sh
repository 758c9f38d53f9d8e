#!/bin/bash
# Runs on the GPU box (gpurun): the device inflater's kernels under rocprofv3 --kernel-trace --stats (tools/inflate_probe.py, 10M reads:
# one gzip member and BGZF through file -> HBM -> inflate -> parse -> hash) -> gpurun_out/prof_inflate/{kernel_stats.csv, probe.json}
cd /tmp; export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$R/gpurun_out/prof_inflate"
rm -rf "$OUT"; mkdir -p "$OUT"
cd "$R"
timeout -s KILL 900 python3 tools/inflate_probe.py ${1:-10000000} > "$OUT/probe.json" 2> "$OUT/probe.err"
timeout -s KILL 900 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o run -- python3 tools/inflate_probe.py ${1:-10000000} > "$OUT/probe_under_rocprof.json" 2> "$OUT/stats.log"
f=$(ls "$OUT"/stats/*/run_kernel_stats.csv "$OUT"/stats/run_kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && grep -h "Name\|k_find\|k_inflate\|k_chain\|k_resolve\|k_crc\|k_make_tails\|k_sketch_reads\|k_read_\|k_gather\|k_count_new\|k_mark_new" "$f" | sed 's/(.*)",/",/' > "$OUT/kernel_stats.csv"
rm -rf "$OUT/stats"
cat "$OUT/kernel_stats.csv"
