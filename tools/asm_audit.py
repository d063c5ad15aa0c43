#!/usr/bin/env python3
"""Audit a kernel's assembly for the inline-asm LDS-load hazard: hipcc counts an asm load's VGPRs as written when the asm
statement ends, so it may copy / read them before the data lands.  The kernels keep every asm `ds_read*` and the asm
`s_waitcnt lgkmcnt` that covers it inside ONE basic block; this tool checks that
  (1) no compiler instruction reads an asm-loaded register before an asm s_waitcnt lgkmcnt follows the load, and
  (2) no asm-loaded register is still un-waited at a label or branch (a copy at a control-flow merge would read it early).
usage: tools/asm_audit.py file.s   (exit status 1 when something is flagged)"""
import re, sys
lines = [l.strip() for l in open(sys.argv[1]).read().split("\n")]
def regs(tok):
    tok = tok.strip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
pending = {}   # reg -> (line, serial) of the asm read
serial = 0
in_asm = False
bad = 0
for i, l in enumerate(lines):
    if l.startswith(";;#ASMSTART"): in_asm = True; continue
    if l.startswith(";;#ASMEND"): in_asm = False; continue
    if not l or l.startswith((";", ".")) and not l.endswith(":"): continue
    if l.endswith(":") or l.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        if pending:
            bad += 1
            print(f"line {i}: {l}   <-- asm-loaded v{sorted(pending)[:8]}... (from line {min(v[0] for v in pending.values())}) not waited for at this control-flow point")
            pending = {}
        continue
    ops = l.split()
    if in_asm and ops[0] == "s_waitcnt" and "lgkmcnt" in l:
        n = int(re.search(r"lgkmcnt\((\d+)\)", l).group(1))
        order = sorted({v[1] for v in pending.values()})
        keep = set(order[len(order) - n:]) if n else set()
        pending = {r: v for r, v in pending.items() if v[1] in keep}
        continue
    if in_asm and ops[0].startswith("ds_read"):
        serial += 1
        for r in regs(ops[1]): pending[r] = (i, serial)
        continue
    if in_asm: continue
    toks = ops[1:]
    is_store = ops[0].startswith(("global_store", "buffer_store", "ds_write", "scratch_store", "global_atomic"))
    srcs = set()
    for k, tok in enumerate(toks):
        if k == 0 and not is_store: continue     # destination
        srcs |= regs(tok)
    hit = srcs & set(pending)
    if hit:
        bad += 1
        print(f"line {i}: {l}   <-- reads v{sorted(hit)} loaded by asm at line {pending[min(hit)][0]} before its wait")
print("flagged:", bad)
sys.exit(1 if bad else 0)
