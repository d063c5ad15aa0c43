#!/usr/bin/env python3
"""Exact-fp32 sharded run, one remote 8192 x 8192 block (rank 0 of 2, driven through the C-ABI on one GPU): the generic forward that saves its
fp32 exponentials + the saved backward (bwd_saved32_kernel<..., RECT>) against the recomputing pair.  usage: time_fp32_rect.py [b] [D]"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crossclr_amd
from crossclr_amd import _native as nat, loss as L
from bench import make_inputs
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib, p, dev = nat.library(), L._ptr, torch.device("cuda")
world = 2
v, t = make_inputs(world * b, D, 3)
v, t = v.cuda(), t.cuda()
plans = [nat.make_plan(b, D, world, r, nat.MODE_FP32) for r in range(world)]
pl = plans[0]
stream = L._stream_for(v)
f32 = dict(dtype=torch.float32, device=dev)
xall = torch.empty(world * pl.operand_bytes, dtype=torch.uint8, device=dev)
inv, diag = [torch.empty(2 * pl.bpad, **f32) for _ in range(world)], [torch.empty(pl.bpad, **f32) for _ in range(world)]
for r in range(world):
    nat.check(lib.crossclr_normalize(ctypes.byref(plans[r]), p(v[r * b:]), p(t[r * b:]), v.stride(0), t.stride(0), nat.IN_F32,
                                     p(xall[r * pl.operand_bytes:]), p(inv[r]), p(diag[r]), stream))
pp, xr = ctypes.byref(pl), xall[:pl.operand_bytes]
part = torch.empty(pl.fwd_ws_floats, **f32)
rz, wrz = torch.rand(world, 2 * pl.bpad, **f32) * 1e-4, torch.rand(world, 2 * pl.bpad, **f32) * 1e-4
st = torch.empty(lib.crossclr_rect_stash_bytes(pp, 1), dtype=torch.uint8, device=dev)
gbuf = torch.zeros(pl.gbuf_bytes // 4, **f32)
stages = {
    "forward (recompute path)": lambda: lib.crossclr_forward(pp, p(xr), p(xall), world, 0, 0, 0.03, 0.8, p(part), pl.fwd_slots, stream),
    "forward_rect_save": lambda: lib.crossclr_forward_rect_save(pp, p(xr), p(xall), 1, 1, 0, 0.03, 0.8, None, p(part), pl.fwd_slots, None, p(st), stream),
    "backward (recompute)": lambda: lib.crossclr_backward(pp, p(xr), p(xall), world, 0, 0, 0.03, 0.8, p(rz[0]), p(wrz[0]), p(rz), p(wrz), p(gbuf), 1, stream),
    "backward_rect_saved": lambda: lib.crossclr_backward_rect_saved(pp, p(xall), p(st), 1, 1, 0.03, 0.8, p(rz[0]), p(wrz[0]), p(rz), p(wrz), None, p(gbuf), 1, stream),
}
for name, fn in stages.items():
    for _ in range(3): nat.check(fn())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
    for a, z in ev:
        a.record(); nat.check(fn()); z.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(z) for a, z in ev)[3]
    print(f"b={b} D={D} fp32, one remote block: {name}: {ms:.3f} ms", flush=True)
