#!/usr/bin/env python3
"""Repeat-launch consistency of the save-for-backward pair through the C-ABI: is the stash the same bytes on every forward,
and is the gradient buffer the same on every backward from one fixed stash?  usage: dbg_stash.py B D"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crossclr_amd import _native as nat, loss as L
B, D = int(sys.argv[1]), int(sys.argv[2])
lib, p = nat.library(), L._ptr
g = torch.Generator().manual_seed(B + D)
v = torch.randn(B, D, generator=g).cuda(); t = torch.randn(B, D, generator=g).cuda()
plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16); pp = ctypes.byref(plan)
st = L._stream_for(v); f32 = dict(dtype=torch.float32, device="cuda")
x = torch.empty(plan.operand_bytes, dtype=torch.uint8, device="cuda"); inv = torch.empty(2 * plan.bpad, **f32); dg = torch.empty(plan.bpad, **f32)
nat.check(lib.crossclr_normalize(pp, p(v), p(t), D, D, nat.IN_F32, p(x), p(inv), p(dg), st))
part = torch.empty(plan.fwd_ws_floats, **f32)
logz, rz, wrz = (torch.empty(2 * plan.bpad, **f32) for _ in range(3))
ls = torch.empty(plan.loss_ws_doubles, dtype=torch.float64, device="cuda")
stashes = []
for i in range(6):
    s = torch.zeros(plan.stash_bytes, dtype=torch.uint8, device="cuda")
    nat.check(lib.crossclr_forward_save(pp, p(x), 0.05, 0.8, None, p(part), 0, p(s), st))
    nat.check(lib.crossclr_forward_finish(pp, p(part), plan.fwd_slots, p(dg), 0.05, 0.8, p(logz), p(rz), p(wrz), p(ls), st))
    torch.cuda.synchronize()
    stashes.append(s)
    if i: print(f"forward {i}: stash bytes differing from forward 0: {(s != stashes[0]).sum().item()} of {plan.stash_bytes}; loss {ls[1].item():.12f}")
gb = []
for i in range(8):
    gbuf = torch.zeros(plan.gbuf_bytes // 4, **f32)
    nat.check(lib.crossclr_backward_saved(pp, p(x), p(stashes[0]), 0.05, 0.8, p(rz), p(wrz), None, p(gbuf), 0, st))
    torch.cuda.synchronize()
    gb.append(gbuf)
    if i:
        d = (gbuf != gb[0])
        idx = d.nonzero().flatten()
        msg = ""
        if idx.numel():
            n2, dp = 2 * plan.bpad, plan.Dpad
            k = idx[:6].tolist()
            msg = " first at (slice,row,col): " + str([(j // (n2 * dp), (j // dp) % n2, j % dp) for j in k]) + f" rows touched: {torch.unique((idx // dp) % n2).numel()}"
        print(f"backward {i}: gbuf elements differing from backward 0: {d.sum().item()}{msg}")
print("plan: bpad", plan.bpad, "Dpad", plan.Dpad, "slices", plan.bwd_slices)
