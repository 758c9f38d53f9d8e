#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel statistics and the HBM / SQ PMC passes over bench.py's workload, then
# tools/summarize_profiles.py turns them into the files committed under profiles/<round>/.
# Usage: bash tools/collect_profiles.sh <tag> [bench args, e.g. --config 1]      (outputs under gpurun_out/prof_<tag>/)
# Every profiled run has ALL kernels on ONE stream (MG_SINGLE_STREAM=1: no two kernels overlap, so a kernel's average
# duration in the trace is its own), no CPU baseline, no secondary workloads; --pmc passes carry --kernel-trace only.
set -u
TAG=${1:-x}; shift
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
ARGS="--steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_kernel_table $*"
timeout -s KILL 600 python3 bench.py --no_secondary "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
MG_SINGLE_STREAM=1 timeout -s KILL 600 python3 bench.py --no_cpu_baseline --no_secondary "$@" > "$OUT/bench_single_stream.json" 2>> "$OUT/bench.err"
# the DEFAULT schedule (stage A of consecutive passes on two alternating streams, stage C on a third): the trace that
# evidences the driver-timed step; kernels of different passes overlap here, so averages are NOT a kernel's own time
timeout -s KILL 900 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats_pipelined" -o run -- python3 bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_secondary --no_kernel_table "$@" > "$OUT/bench_pipelined_under_rocprof.json" 2> "$OUT/stats_pipelined.log"
python3 tools/trace_timeline.py "$OUT/stats_pipelined" "$OUT/pipelined_timeline.json" > "$OUT/pipelined_timeline.txt" 2>&1
export MG_SINGLE_STREAM=1
# (30 passes: the first launches of a process run at low clocks — with 5 passes they were a third of the average)
timeout -s KILL 900 rocprofv3 --output-format csv --kernel-trace --stats -d "$OUT/stats" -o run -- python3 bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_secondary --no_kernel_table "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 900 rocprofv3 --output-format csv --pmc $c --kernel-trace -d "$OUT/pmc_$c" -o run -- python3 bench.py $ARGS > "$OUT/pmc_$c.log" 2>&1
done
timeout -s KILL 900 rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
  --kernel-trace -d "$OUT/pmc_SQ" -o run -- python3 bench.py $ARGS > "$OUT/pmc_SQ.log" 2>&1
python3 tools/summarize_profiles.py "$OUT" > "$OUT/summary.log" 2>&1
# the per-opcode pricing of the fused kernel's hot loop, from the assembly of the library as built here (MG_PROFILE_ASM=1: a
# minute of compiling on the GPU box; the one-k kernel of the reference pipeline — mg_sketch.hip, three minutes — is priced in
# the build container instead: tools/valu_roofline.py --kernel 14k_sketch_readsILi51ELi0E --ks 51 --positions_per_iteration 2)
if [ "${MG_PROFILE_ASM:-0}" = "1" ]; then
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 --cuda-device-only -S -o "$OUT/mg_sketch_multi.s" metalign_amd/csrc/mg_sketch_multi.hip 2>/dev/null
CLASSES=$(ls profiles/*/valu_classes.json | tail -1)
python3 tools/valu_roofline.py "$OUT/mg_sketch_multi.s" "$CLASSES" --out "$OUT/k1_valu_roofline.json" > "$OUT/k1_valu_roofline.txt" 2>&1
rm -f "$OUT/mg_sketch_multi.s"
fi
for d in stats stats_pipelined pmc_FETCH_SIZE pmc_WRITE_SIZE pmc_SQ; do rm -rf "$OUT/$d"; done
tail -n 30 "$OUT/summary.log"
