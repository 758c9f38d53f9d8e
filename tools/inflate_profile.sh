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
# the decoder's instruction mix (a pass per counter set; --pmc with --kernel-trace only)
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 900 rocprofv3 --output-format csv --pmc $set --kernel-trace -d "$OUT/pmc_$tag" -o run -- python3 tools/inflate_probe.py ${1:-10000000} > /dev/null 2> "$OUT/pmc_$tag.log"
  f=$(ls "$OUT"/pmc_$tag/*/run_counter_collection.csv "$OUT"/pmc_$tag/run_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" >> "$OUT/pmc_summary.txt" <<'PY'
import csv, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if not any(x in k for x in ("k_inflate", "k_find_block", "k_resolve_text", "k_chain", "k_crc_seg")): continue
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k in sorted(tot):
    print(k, "dispatches", len(n[k]), {c: int(v) for c, v in tot[k].items()})
PY
  rm -rf "$OUT/pmc_$tag"
done
cat "$OUT/pmc_summary.txt"
cat "$OUT/kernel_stats.csv"
