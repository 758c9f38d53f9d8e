"""GPU-side timeline of one bench step from a rocprofv3 kernel + memory-copy trace: per kernel, the idle gap before
it and its duration.  Usage (on the GPU box):
  rocprofv3 --output-format csv --kernel-trace --memory-copy-trace -d /tmp/st -o run -- python3 bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_kernel_table
  python3 tools/gpu_gaps.py /tmp/st"""
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-34:]) for r in csv.DictReader(open(f))]
for m in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(m)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
idx = [i for i, r in enumerate(rows) if "pass_prepare" in r[2] or "acc_reset" in r[2]]
a, b = idx[-2], idx[-1]
prev, tot_gap, tot_busy = None, 0.0, 0.0
for s, e, n in rows[a:b]:
    gap = (s - prev) / 1000 if prev else 0.0
    tot_gap += max(gap, 0.0)
    tot_busy += (e - s) / 1000
    print("%8.1f gap %8.1f dur  %s" % (gap, (e - s) / 1000, n))
    prev = max(e, prev or 0)
print("step span %.1f us, busy %.1f, gaps %.1f" % ((rows[b][0] - rows[a][0]) / 1000, tot_busy, tot_gap))
