#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv: mean per dispatch per kernel (short names)."""
import csv, collections, sys, re
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void crossclr::", "")[:40]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        if not any(x in k for x in ("fwd", "bwd", "normalize")): continue
        print(f"{k:42s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(cs.items())))
