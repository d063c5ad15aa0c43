#!/usr/bin/env python3
"""Fillers between consecutive MFMAs of every loop of a kernel's assembly: tools/slot_hist.py file.s
(S scalar, L LDS, M vector memory, V VALU, W s_waitcnt, B barrier, N s_nop)."""
import re, sys
lines = []
for l in open(sys.argv[1]):
    l = l.split(";")[0].strip()
    if not l or (l.startswith(".") and not l.endswith(":")): continue
    lines.append(l)
labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
def cls(op):
    if op.startswith("s_waitcnt"): return "W"
    if op.startswith("s_barrier"): return "B"
    if op.startswith("s_nop"): return "N"
    if op.startswith("s_"): return "S"
    if op.startswith("ds_"): return "L"
    if op.startswith(("buffer_", "global_")): return "M"
    return "V"
for i, l in enumerate(lines):
    m = re.match(r"s_cbranch_\S+ (\S+)", l)
    if not (m and m.group(1) in labels and labels[m.group(1)] < i): continue
    a = labels[m.group(1)]
    body = [x for x in lines[a:i + 1] if not x.endswith(":")]
    n = sum(1 for x in body if x.startswith("v_mfma"))
    if n < 8: continue
    print(f"loop {m.group(1)}: {len(body)} instructions, {n} mfma")
    cur = ""
    k = 0
    for x in body:
        op = x.split()[0]
        if op.startswith("v_mfma"):
            print(f"  {k:2d} {len(cur):2d} {cur}")
            cur = ""; k += 1
        else: cur += cls(op)
    print(f"  end {len(cur):2d} {cur}")
