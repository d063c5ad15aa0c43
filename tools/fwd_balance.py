#!/usr/bin/env python3
"""What does a masked (diagonal) tile and a row-block switch cost the symmetric forward, in plain tiles?  Per-block main-loop times of one
launch (library built with -DCROSSCLR_TIMING, see tools/timeline.py) regressed on each range's composition: plain tiles, masked tiles,
row-block (segment) starts.  The work list below is a Python port of csrc/crossclr_device.h (fwdw_* / fwd_make_work / fwd_make_perm) for the
cost set the library was built with (CROSSCLR_FWD_COST_FIRST / _MASKED / _PLAIN, default 3 / 2 / 1).
    CROSSCLR_HIP_LIBRARY=variants/libtm.so python tools/fwd_balance.py [B] [D] [first masked plain]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from crossclr_amd import _native as nat, loss as L
from oracle import crossclr_oracle as orc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
C0, CM, CP = (int(x) for x in sys.argv[3:6]) if len(sys.argv) > 5 else (3, 2, 1)
lib = nat.library()
lib.crossclr_debug_timing.restype = ctypes.c_int
lib.crossclr_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
v, t = orc.make_inputs("randn", B, D, 1234)
v, t = v.cuda(), t.cuda()
_, ws = L._forward_impl(v, t, 0.03, 0.8, "bf16", None, None, None, save_for_backward=True)
plan, pp, p = ws.plan, ctypes.byref(ws.plan), L._ptr
stream = L._stream_for(v)
part = torch.empty(plan.fwd_ws_floats, dtype=torch.float32, device="cuda")
fn = lambda: lib.crossclr_forward_save(pp, p(ws.xhat), 0.03, 0.8, None, p(part), 0, p(ws.stash), stream)

tpr = 8 if plan.Dpad <= 512 else 4
NT = 2 * plan.bpad // 32
NB = NT // tpr
items = lambda rb: NT - tpr * rb
cost = lambda j: C0 if j == 0 else (CM if j < tpr else CP)
# units: explicit prefix sums (the C++ closed forms give the same numbers)
unit_of_item = [[0] * (items(rb) + 1) for rb in range(NB)]
row_units = []
for rb in range(NB):
    u = 0
    for j in range(items(rb)):
        unit_of_item[rb][j] = u
        u += cost(j)
    unit_of_item[rb][items(rb)] = u
    row_units.append(u)
unit_prefix = np.concatenate([[0], np.cumsum(row_units)])
units = int(unit_prefix[-1])
min_per = max(C0, 2)
nb = min(units // min_per, plan.fwd_blocks)
per = max((units + nb - 1) // nb, min_per)
nblk = (units + per - 1) // per
flat = []                                   # (row block, item) in list order with the item's first unit
for rb in range(NB):
    for j in range(items(rb)):
        flat.append((rb, j, int(unit_prefix[rb]) + unit_of_item[rb][j]))
first_unit = np.array([f[2] for f in flat])
begin = [int(np.searchsorted(first_unit, c * per, side="left")) for c in range(nblk)] + [len(flat)]
comp = np.zeros((nblk, 3))                  # plain, masked, segment starts
key = np.zeros(nblk, dtype=np.int64)
for c in range(nblk):
    seg = -1
    for rb, j, _ in flat[begin[c]:begin[c + 1]]:
        comp[c, 0 if j >= tpr else 1] += 1
        if rb != seg:
            comp[c, 2] += 1
            seg = rb
    rb, j, _ = flat[begin[c]] if begin[c] < len(flat) else (0, 1 << 30, 0)
    key[c] = tpr * rb + j
order = sorted(range(nblk), key=lambda c: (key[c], c))
perm = list(range(256))
if 16 <= nblk <= 256:
    off = 0
    for x in range(8):
        cnt = (nblk - x + 7) // 8
        for i in range(cnt):
            perm[x + 8 * i] = order[off + i]
        off += cnt

for _ in range(400):
    nat.check(fn())
torch.cuda.synchronize()
samples = []
for rep in range(5):
    nat.check(fn()); torch.cuda.synchronize()
    raw = np.zeros((256, 8), dtype=np.uint64)
    nat.check(lib.crossclr_debug_timing(raw.ctypes.data, 256))
    m = raw[:nblk, :4].astype(np.int64)
    samples.append(((m[:, 2] - m[:, 1]) * 0.01, (m[:, 3] - m[:, 0]) * 0.01, raw[:nblk, 7] & 0xF))
loop = np.median(np.stack([s[0] for s in samples]), axis=0)       # us, per BLOCK
life = np.median(np.stack([s[1] for s in samples]), axis=0)
X = np.stack([comp[perm[b]] for b in range(nblk)])
coef, *_ = np.linalg.lstsq(X, loop, rcond=None)
print(f"B={B} D={D} built-in costs first/masked/plain = {C0}/{CM}/{CP}: {nblk} ranges of {per} units ({units} units)")
print(f"main loop per block: median {np.median(loop):.1f} us, min {loop.min():.1f}, max {loop.max():.1f}; lifetime max {life.max():.1f}")
print(f"least squares: plain tile {coef[0]:.3f} us, masked tile {coef[1]:.3f} us (= {coef[1] / coef[0]:.2f} plain), row-block start {coef[2]:.3f} us (= {coef[2] / coef[0]:.2f} plain); "
      f"residual rms {np.sqrt(np.mean((X @ coef - loop) ** 2)):.2f} us")
worst = np.argsort(loop)[-5:]
print("slowest blocks (block: range plain/masked/starts -> us): " + ", ".join(f"{b}: {perm[b]} {int(X[b,0])}/{int(X[b,1])}/{int(X[b,2])} -> {loop[b]:.1f}" for b in worst))
best = np.argsort(loop)[:5]
print("fastest blocks: " + ", ".join(f"{b}: {perm[b]} {int(X[b,0])}/{int(X[b,1])}/{int(X[b,2])} -> {loop[b]:.1f}" for b in best))
