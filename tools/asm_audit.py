#!/usr/bin/env python3
"""Audit a kernel's assembly for the inline-asm LDS-load hazard: hipcc counts an asm load's VGPRs as written when the asm
statement ends, so it may copy / read them before the data lands.  The kernels keep every asm `ds_read*` and the asm
`s_waitcnt lgkmcnt` that covers it inside ONE basic block; this tool checks that
  (1) no compiler instruction reads an asm-loaded register before an asm s_waitcnt lgkmcnt follows the load, and
  (2) no asm-loaded register is still un-waited at a label or branch (a copy at a control-flow merge would read it early).
The same two checks for the asm `buffer_load_dwordx4` into VGPRs of the fragment-major backward, against `s_waitcnt vmcnt(N)`
(every VMEM instruction between the load and the wait counts: vmcnt retires in order).
usage: tools/asm_audit.py file.s   (exit status 1 when something is flagged)"""
import re, sys
lines = [l.strip() for l in open(sys.argv[1]).read().split("\n")]
def regs(tok):
    tok = tok.strip(",")
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
pending = {}   # reg -> (line, serial) of the asm LDS read
vpending = {}  # reg -> (line, VMEM serial) of an asm buffer load into VGPRs (the fragment-major backward's B fragments)
serial = 0
vserial = 0    # every VMEM instruction, asm or compiler's (vmcnt retires loads AND stores in order on gfx9)
in_asm = False
bad = 0
VMEM = ("buffer_load", "buffer_store", "buffer_atomic", "global_load", "global_store", "global_atomic", "flat_load", "flat_store",
        "scratch_load", "scratch_store")
for i, l in enumerate(lines):
    if l.startswith(";;#ASMSTART"): in_asm = True; continue
    if l.startswith(";;#ASMEND"): in_asm = False; continue
    if not l or l.startswith((";", ".")) and not l.endswith(":"): continue
    if l.endswith(":") or l.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
        if pending or vpending:
            bad += 1
            both = dict(pending); both.update(vpending)
            print(f"line {i}: {l}   <-- asm-loaded v{sorted(both)[:8]}... (from line {min(v[0] for v in both.values())}) not waited for at this control-flow point")
            pending = {}; vpending = {}
        continue
    ops = l.split()
    if ops[0] == "s_waitcnt" and "vmcnt" in l:     # (the compiler's own waits count too: they can only retire MORE)
        n = int(re.search(r"vmcnt\((\d+)\)", l).group(1))
        vpending = {r: v for r, v in vpending.items() if v[1] > vserial - n}
    if in_asm and ops[0] == "s_waitcnt" and "lgkmcnt" in l:
        n = int(re.search(r"lgkmcnt\((\d+)\)", l).group(1))
        order = sorted({v[1] for v in pending.values()})
        keep = set(order[len(order) - n:]) if n else set()
        pending = {r: v for r, v in pending.items() if v[1] in keep}
        continue
    if ops[0] == "s_waitcnt": continue
    if ops[0].startswith(VMEM):
        vserial += 1
        if in_asm and ops[0].startswith("buffer_load") and " lds" not in l:
            for r in regs(ops[1]): vpending[r] = (i, vserial)
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
    hit = srcs & (set(pending) | set(vpending))
    if hit:
        bad += 1
        src_line = (pending.get(min(hit)) or vpending.get(min(hit)))[0]
        print(f"line {i}: {l}   <-- reads v{sorted(hit)} loaded by asm at line {src_line} before its wait")
print("flagged:", bad)
sys.exit(1 if bad else 0)
