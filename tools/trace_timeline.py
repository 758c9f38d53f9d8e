"""GPU-side timeline of the DEFAULT (pipelined, multi-stream) schedule from a rocprofv3 kernel trace: the period between
consecutive stage-A launches (= the pass rate the device sustains, to hold against bench.py's ms_per_step), how long each
stage-A launch ran, how much of it overlapped the previous pass's launch, and one steady-state pass kernel by kernel
(start offset from the pass's stage-A launch, duration, queue/stream id).
Usage (on the GPU box):
  rocprofv3 --output-format csv --kernel-trace --stats -d D -o run -- python3 bench.py --steps 20 --warmup 3 --no_cpu_baseline --no_secondary --no_kernel_table
  python3 tools/trace_timeline.py D [json-out]"""
import csv
import glob
import json
import re
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    m = re.search(r"mg::(k_\w+)", r["Kernel_Name"].replace("(anonymous namespace)::", ""))
    name = m.group(1) if m else r["Kernel_Name"].split("(")[0][-40:]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
a = [r for r in rows if r[2].startswith("k_count_kmers")] or [r for r in rows if r[2].startswith("k_sketch_reads")]  # (stage A: by k-mer identity, or the read sketch)
tail = a[-20:]  # the timed region's launches (bench.py --steps 20)
starts = [r[0] for r in tail]
periods = [(y - x) / 1e6 for x, y in zip(starts, starts[1:])]
durs = [(r[1] - r[0]) / 1e6 for r in tail]
overlap = [max(0, min(p[1], q[1]) - q[0]) / 1e6 for p, q in zip(tail, tail[1:])]
res = {"stage_a_launches_in_trace": len(a), "last_20": {
    "period_ms_mean": sum(periods) / len(periods), "period_ms_min": min(periods), "period_ms_max": max(periods),
    "launch_ms_mean": sum(durs) / len(durs), "launch_ms_min": min(durs), "launch_ms_max": max(durs),
    "overlap_with_previous_launch_ms_mean": sum(overlap) / len(overlap),
    "span_ms": (tail[-1][1] - tail[0][0]) / 1e6}}
print(json.dumps(res, indent=1))
# one steady-state pass: everything that starts between two consecutive stage-A starts in the middle of the tail
lo, hi = tail[9][0], tail[10][0]
print("\none steady-state pass (offset from its stage-A start, us | duration, us | stream | kernel):")
for s, e, n, q in rows:
    if lo <= s < hi:
        print("%10.1f %10.1f  s%-4s %s" % ((s - lo) / 1e3, (e - s) / 1e3, q, n))
busy = sorted((s, e) for s, e, _, _ in rows if e > lo and s < hi)
cur, idle = lo, 0
for s, e in busy:
    if s > cur:
        idle += s - cur
    cur = max(cur, min(e, hi))
print("window %.1f us, no kernel resident for %.1f us" % ((hi - lo) / 1e3, idle / 1e3))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
