#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel statistics and the two HBM PMC passes over bench.py's default
# workload, then tools/summarize_profiles.py turns them into the files committed under profiles/<round>/.
# Usage: bash tools/collect_profiles.sh <tag>        (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-x}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
CMD="python3 bench.py --steps 5 --warmup 2 --no_cpu_baseline"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o run -- $CMD > "$OUT/stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --output-format csv --pmc $c --kernel-trace -d "$OUT/pmc_$c" -o run -- $CMD > "$OUT/pmc_$c.log" 2>&1
done
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace -d "$OUT/pmc_SQ" -o run -- $CMD > "$OUT/pmc_SQ.log" 2>&1
python3 tools/summarize_profiles.py "$OUT" > "$OUT/summary.log" 2>&1
du -sh "$OUT"/* | tail -12
# keep the condensed files and the (small) per-kernel CSVs; the raw traces stay on the box
mkdir -p "$OUT/raw"
for d in stats pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do
  for f in $(find "$OUT/$d" -name '*kernel_stats.csv' -o -name '*counter_collection.csv' 2>/dev/null); do
    sz=$(stat -c %s "$f"); if [ "$sz" -lt 4000000 ]; then cp "$f" "$OUT/raw/${d}_$(basename "$f")"; fi
  done
  rm -rf "$OUT/$d"
done
tail -n 3 "$OUT/stats.log"
tail -n 40 "$OUT/summary.log"
