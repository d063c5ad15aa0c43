#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd SQLite result (gpurun_out/<dir>/*/*_results.db) into the text summary
committed under profiles/.  usage: summarize_rocpd.py <db> [title]"""
import sqlite3
import sys

db, title = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
con = sqlite3.connect(db)
print(f"# rocprofv3 --kernel-trace --stats  {title}")
print(f"# source: {db}")
print(f"{'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name[:90]:90s} {calls:6d} {total:12.1f} {avg:10.2f} {pct:6.2f}")
