#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV (one row per dispatch) -> per-kernel summary with the statistics bench.py's HIP events can be
compared with: calls, average over ALL dispatches (what --stats prints), and average / median / min / max over the SETTLED
dispatches = the last `--settled-frac` (default half) of each kernel's dispatches in time order (bench.py runs 40 settle steps
before it measures; the first dispatches of a process run at a ramping clock).
usage: tools/kernel_trace_summary.py <..._kernel_trace.csv> [--settled-frac 0.5] > profiles/rNN_kernel_stats.csv"""
import csv, statistics, sys
path = sys.argv[1]
frac = float(sys.argv[sys.argv.index("--settled-frac") + 1]) if "--settled-frac" in sys.argv else 0.5
rows = {}
with open(path) as f:
    for r in csv.DictReader(f):
        name = r.get("Kernel_Name") or r.get("kernel_name")
        t0, t1 = int(r.get("Start_Timestamp") or r.get("start")), int(r.get("End_Timestamp") or r.get("end"))
        rows.setdefault(name, []).append((t0, (t1 - t0) / 1e3))
total = sum(d for v in rows.values() for _, d in v)
w = csv.writer(sys.stdout)
w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs_all", "Percentage", "SettledCalls", "SettledAverageUs", "SettledMedianUs",
            "SettledMinUs", "SettledMaxUs", "SettledStdDevUs"])
for name, v in sorted(rows.items(), key=lambda kv: -sum(d for _, d in kv[1])):
    v.sort()
    d = [x for _, x in v]
    s = d[int(len(d) * (1 - frac)):] or d
    w.writerow([name, len(d), round(sum(d), 2), round(sum(d) / len(d), 3), round(100 * sum(d) / total, 2), len(s), round(sum(s) / len(s), 3),
                round(statistics.median(s), 3), round(min(s), 3), round(max(s), 3), round(statistics.pstdev(s), 3)])
