#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: short kernel name, calls, average / min / max microseconds, share.  usage: kstats.py FILE [N]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[:n]:
    name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "").replace("crossclr::", "")
    name = re.sub(r"at::native::", "", name)[:90]
    print(f"{name:90s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:9.2f} us  min {float(r['MinNs'])/1e3:9.2f}  max {float(r['MaxNs'])/1e3:9.2f}  {float(r['Percentage']):5.2f} %")
