#!/usr/bin/env python3
"""Compute time of ONE rank of an N-rank run of a plan on the GENERIC sharded branch (wide bf16 plans, D > 1024; exact-fp32 plans), measured
on one GPU, communication excluded: every rank evaluates its rows against all other ranks (no pair scheme there), local block + ONE
rectangular launch over the other world - 1 ranks, with the saved exponentials (what the module runs) and with the recomputing backward
beside it.  Projects the weak-scaling metric B_global^2 / t.  usage: shard_bench_generic.py [b] [D] [bf16|fp32]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from crossclr_amd import _native as nat, loss as L
from oracle import crossclr_oracle as orc
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
mode = nat.MODE_FP32 if (len(sys.argv) > 3 and sys.argv[3] == "fp32") else nat.MODE_BF16
lib, p = nat.library(), L._ptr
v, t = orc.make_inputs("randn", b, D, 1234)
v, t = v.cuda(), t.cuda()
stream = L._stream_for(v)
f32 = dict(dtype=torch.float32, device="cuda")
T, W = 0.03, 0.8


def med(fn, n=5, warm=2):
    for _ in range(warm): nat.check(fn())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, z in ev:
        a.record(); nat.check(fn()); z.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(z) for a, z in ev)[n // 2]


# settle: a freshly acquired GPU runs its first steps ~9 % slower (bench.py does the same)
_crit = crossclr_amd.CrossCLR_onlyIntraModality(T, W, compute_mode="fp32" if mode == nat.MODE_FP32 else "bf16").cuda()
_v, _t = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
for _ in range(40):
    _v.grad = _t.grad = None
    _crit(_v, _t).backward()
torch.cuda.synchronize()
del _crit, _v, _t
base = None
for world in (1, 2, 4, 8):
    rank = world // 2
    plan = nat.make_plan(b, D, world, rank, mode)
    pp = ctypes.byref(plan)
    assert plan.fast_path == 0 and plan.stash_bytes > 0, "a plan of the generic branch with a save-for-backward path (D > 1024 bf16, or fp32)"
    xall = torch.empty(world * plan.operand_bytes, dtype=torch.uint8, device="cuda")
    inv, diag = torch.empty(2 * plan.bpad, **f32), torch.empty(plan.bpad, **f32)
    for r in range(world):   # every "rank" holds the same rows: values do not matter for timing
        nat.check(lib.crossclr_normalize(pp, p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(xall[r * plan.operand_bytes:]), p(inv), p(diag), stream))
    xr = xall[rank * plan.operand_bytes:(rank + 1) * plan.operand_bytes]
    part = torch.empty(plan.fwd_ws_floats, **f32)
    logz, rz, wrz = (torch.empty(2 * plan.bpad, **f32) for _ in range(3))
    ls = torch.empty(plan.loss_ws_doubles, dtype=torch.float64, device="cuda")
    gbuf = torch.empty(plan.gbuf_bytes // 4, **f32)
    go = torch.ones(1, dtype=torch.float64, device="cuda")
    gv, gt = torch.empty_like(v), torch.empty_like(t)
    stash = torch.empty(plan.stash_bytes, dtype=torch.uint8, device="cuda")
    st = {}
    st["normalize"] = med(lambda: lib.crossclr_normalize(pp, p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(xr), p(inv), p(diag), stream))
    st["fwd_local(save)"] = med(lambda: lib.crossclr_forward_save(pp, p(xr), T, W, None, p(part), 0, p(stash), stream))
    first = (rank + 1) % world
    if world > 1:
        nb = lib.crossclr_rect_stash_bytes(pp, world - 1)
        assert nb > 0
        st_r = torch.empty(nb, dtype=torch.uint8, device="cuda")
        st["fwd_remote(save)"] = med(lambda: lib.crossclr_forward_rect_save(pp, p(xr), p(xall), first, world - 1, 0, T, W, None, p(part), plan.fwd_slots, None, p(st_r), stream))
        st["[fwd_remote, no save]"] = med(lambda: lib.crossclr_forward(pp, p(xr), p(xall), world, 0, rank, T, W, p(part), plan.fwd_slots, stream), n=3, warm=1)
    st["fwd_finish"] = med(lambda: lib.crossclr_forward_finish(pp, p(part), (2 if world > 1 else 1) * plan.fwd_slots, p(diag), T, W, p(logz), p(rz), p(wrz), p(ls), stream))
    rzc, wrzc = rz.repeat(world), wrz.repeat(world)
    st["bwd_local(saved)"] = med(lambda: lib.crossclr_backward_saved(pp, p(xr), p(stash), T, W, p(rz), p(wrz), None, p(gbuf), 0, stream))
    if world > 1:
        st["bwd_remote(saved)"] = med(lambda: lib.crossclr_backward_rect_saved(pp, p(xall), p(st_r), first, world - 1, T, W, p(rz), p(wrz), p(rzc), p(wrzc), None, p(gbuf), 1, stream))
        st["[bwd_remote, recompute]"] = med(lambda: lib.crossclr_backward(pp, p(xr), p(xall), world, 0, rank, T, W, p(rz), p(wrz), p(rzc), p(wrzc), p(gbuf), 1, stream), n=3, warm=1)
    st["bwd_finish"] = med(lambda: lib.crossclr_backward_finish(pp, p(gbuf), p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(inv), T, p(go), p(gv), p(gt), gv.stride(0), gt.stride(0), stream))
    tot = sum(x for k, x in st.items() if not k.startswith("["))
    tot_re = tot - st.get("fwd_remote(save)", 0) - st.get("bwd_remote(saved)", 0) + st.get("[fwd_remote, no save]", 0) + st.get("[bwd_remote, recompute]", 0)
    rate, rate_re = (world * b) ** 2 / tot, (world * b) ** 2 / tot_re
    if base is None: base = rate
    print(f"N={world} b={b} D={D} {'fp32' if mode == nat.MODE_FP32 else 'bf16'}: " + " ".join(f"{k}={x:.3f}" for k, x in st.items()) +
          f" | per-rank compute {tot:.3f} ms = {rate / base:.2f} x the N=1 rate (recomputing remote blocks: {tot_re:.3f} ms = {rate_re / base:.2f} x)", flush=True)
    del stash, xall
    if world > 1: del st_r
    torch.cuda.empty_cache()
