#!/usr/bin/env python3
"""Compute time of ONE rank of an N-rank run, measured on one GPU (no communication): the kernel sequence of the
sharded step (local block, then the gathered operand with skip_rank) against a synthetic gathered operand.
Projects the weak-scaling metric B_global^2 / t for N = 1, 2, 4, 8 at b rows per rank (communication excluded).
usage: shard_bench.py [b] [D]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from crossclr_amd import _native as nat, loss as L
from oracle import crossclr_oracle as orc
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib, p = nat.library(), L._ptr
v, t = orc.make_inputs("randn", b, D, 1234)
v, t = v.cuda(), t.cuda()
stream = L._stream_for(v)
f32 = dict(dtype=torch.float32, device="cuda")
# settle: a freshly acquired GPU runs its first steps ~9 % slower (bench.py does the same)
_crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
_v, _t = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
for _ in range(60):
    _v.grad = _t.grad = None
    _crit(_v, _t).backward()
torch.cuda.synchronize()
base = None
for world in (1, 2, 4, 8):
    rank = world // 2
    plan = nat.make_plan(b, D, world, rank, nat.MODE_BF16)
    pp = ctypes.byref(plan)
    xall = torch.empty(world * plan.operand_bytes, dtype=torch.uint8, device="cuda")
    inv, diag = torch.empty(2 * plan.bpad, **f32), torch.empty(plan.bpad, **f32)
    for r in range(world):   # every "rank" holds the same rows: values do not matter for timing
        nat.check(lib.crossclr_normalize(pp, p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32,
                                         p(xall[r * plan.operand_bytes:]), p(inv), p(diag), stream))
    xr = xall[rank * plan.operand_bytes:(rank + 1) * plan.operand_bytes]
    part = torch.empty(plan.fwd_ws_floats, **f32)
    logz, rz, wrz = (torch.empty(2 * plan.bpad, **f32) for _ in range(3))
    rzc, wrzc = torch.empty(world * 2 * plan.bpad, **f32), torch.empty(world * 2 * plan.bpad, **f32)
    ls = torch.empty(plan.loss_ws_doubles, dtype=torch.float64, device="cuda")
    gbuf = torch.empty(plan.gbuf_bytes // 4, **f32)
    go = torch.ones(1, dtype=torch.float64, device="cuda")
    gv, gt = torch.empty_like(v), torch.empty_like(t)
    stages = {}
    def run(name, fn):
        for _ in range(2): nat.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): nat.check(fn())
        e1.record(); torch.cuda.synchronize()
        stages[name] = e0.elapsed_time(e1) / 5
    stash = torch.empty(plan.stash_bytes, dtype=torch.uint8, device="cuda") if plan.stash_bytes else None
    # the local block takes the fragment-major pair where the module does (loss._use_xf)
    xf = torch.empty(plan.xf_bytes, dtype=torch.uint8, device="cuda") if (stash is not None and plan.xf_bytes and L._use_xf(plan)) else None
    if xf is not None:
        run("normalize(xf)", lambda: lib.crossclr_normalize_xf(pp, p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(xr), p(xf), p(inv), p(diag), stream))
    else:
        run("normalize", lambda: lib.crossclr_normalize(pp, p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(xr), p(inv), p(diag), stream))
    if stash is not None:   # the local block takes the save-for-backward pair, like the module does when a backward follows
        run("fwd_local(save)", lambda: lib.crossclr_forward_save(pp, p(xr), 0.03, 0.8, None, p(part), 0, p(stash), stream))
    else:
        run("fwd_local", lambda: lib.crossclr_forward(pp, p(xr), p(xr), 1, rank, -1, 0.03, 0.8, p(part), 0, stream))
    K = (world - 1) // 2 if world > 2 else 0
    saved = []     # (first_rank, nranks, stash): the remote blocks this rank evaluates itself, exponentials saved like the module does
    if K:
        colsum = torch.empty(K * 2 * plan.bpad, **f32)
        st_p = torch.empty(lib.crossclr_rect_stash_bytes(pp, K), dtype=torch.uint8, device="cuda")
        run("fwd_pairs(save)", lambda: lib.crossclr_forward_rect_save(pp, p(xr), p(xall), (rank + 1) % world, K, 1, 0.03, 0.8, None, p(part),
                                                                    plan.fwd_slots, p(colsum), p(st_p), stream))
        saved.append(((rank + 1) % world, K, st_p))
    if world > 1 and world % 2 == 0:
        opp = (rank + world // 2) % world
        st_a = torch.empty(lib.crossclr_rect_stash_bytes(pp, 1), dtype=torch.uint8, device="cuda")
        run("fwd_antipode(save)", lambda: lib.crossclr_forward_rect_save(pp, p(xr), p(xall), opp, 1, 0, 0.03, 0.8, None, p(part),
                                                                       (2 if K else 1) * plan.fwd_slots, None, p(st_a), stream))
        saved.append((opp, 1, st_a))
    elif K:
        nat.check(lib.crossclr_forward_add(pp, p(part), 2 * plan.fwd_slots, None, stream))
    if K:
        run("fwd_add", lambda: lib.crossclr_forward_add(pp, p(part), 3 * plan.fwd_slots, p(colsum), stream))
        n = 4 * plan.fwd_slots
    elif world > 1:
        n = 2 * plan.fwd_slots
    else:
        n = plan.fwd_slots
    run("fwd_finish", lambda: lib.crossclr_forward_finish(pp, p(part), n, p(diag), 0.03, 0.8, p(logz), p(rz), p(wrz), p(ls), stream))
    for r in range(world):
        rzc[r * 2 * plan.bpad:(r + 1) * 2 * plan.bpad] = rz
        wrzc[r * 2 * plan.bpad:(r + 1) * 2 * plan.bpad] = wrz
    # the module's default: the pair kernel (fast_bwd_xfp_kernel) for the local block AND, on fragment-major copies, for the remote blocks
    # and partner gradients (CROSSCLR_REMOTE_XFP=0: the LDS-staged kernels for the remote ones, as in round 3)
    remote_xfp = xf is not None and os.environ.get("CROSSCLR_REMOTE_XFP", "1") != "0" and world * plan.operand_bytes < (1 << 32)
    xfall = None
    if remote_xfp and saved:
        xfall = torch.empty(world * plan.operand_bytes, dtype=torch.uint8, device="cuda")
        def relay():
            rc = 0
            for first, nr, _ in saved:
                runs = [(first, min(nr, world - first))] + ([(0, nr - (world - first))] if first + nr > world else [])
                for r0, cnt in runs:
                    rc |= lib.crossclr_pack_xf_from_packed(pp, xall.data_ptr() + r0 * plan.operand_bytes, cnt, xfall.data_ptr() + r0 * plan.operand_bytes, stream)
            return rc
        run("xf_from_packed", relay)
    if xf is not None:
        run("bwd_local(saved, xfp)", lambda: lib.crossclr_backward_saved_xfp(pp, p(xf), p(stash), 0.03, 0.8, p(rz), p(wrz), None, p(gbuf), 0, stream))
    elif stash is not None:
        run("bwd_local(saved)", lambda: lib.crossclr_backward_saved(pp, p(xr), p(stash), 0.03, 0.8, p(rz), p(wrz), None, p(gbuf), 0, stream))
    else:
        run("bwd_local", lambda: lib.crossclr_backward(pp, p(xr), p(xr), 1, rank, -1, 0.03, 0.8, p(rz), p(wrz), p(rz), p(wrz), p(gbuf), 0, stream))
    for i, (first, nr, st) in enumerate(saved):
        if xfall is not None:
            run(f"bwd_saved_xfp[{nr}]" + ("" if i == 0 else "'"), lambda: lib.crossclr_backward_rect_saved_xfp(pp, p(xfall), p(st), first, nr, 0.03, 0.8, p(rz), p(wrz), p(rzc),
                                                                                                               p(wrzc), None, p(gbuf), 1, stream))
        else:
            run(f"bwd_saved[{nr}]" + ("" if i == 0 else "'"), lambda: lib.crossclr_backward_rect_saved(pp, p(xall), p(st), first, nr, 0.03, 0.8, p(rz), p(wrz), p(rzc), p(wrzc),
                                                                                                       None, p(gbuf), 1, stream))
    if K and os.environ.get("CROSSCLR_PARTNER_GRADS", "1") != "0":
        # partner gradients (the default): this rank forms the transposed contribution of each of its K pair blocks for the partner
        # (crossclr_backward_rect_saved_t), sums the column slices and -- after the exchange, not timed here -- adds what it received
        tmp = torch.empty(plan.gbuf_bytes // 4, **f32)
        nel = 2 * plan.bpad * plan.Dpad
        outg = torch.empty(K, nel, **f32)
        def partner():
            rc = 0
            for k in range(K):
                if xfall is not None:
                    rc |= lib.crossclr_backward_rect_saved_t_xfp(pp, p(xf), p(st_p), (rank + 1) % world, K, k, 0.03, 0.8, p(rz), p(wrz), p(rzc), p(wrzc), None, p(tmp), stream)
                else:
                    rc |= lib.crossclr_backward_rect_saved_t(pp, p(xr), p(st_p), (rank + 1) % world, K, k, 0.03, 0.8, p(rz), p(wrz), p(rzc), p(wrzc), None, p(tmp), stream)
                torch.sum(tmp.view(-1, nel), 0, out=outg[k])
            for k in range(K):
                gbuf[:nel] += outg[k]
            return rc
        run(f"bwd_partner[{K}] (+slice sums, +adds)", partner)
    elif K:
        run(f"bwd_recompute[{K}]", lambda: lib.crossclr_backward_ranks(pp, p(xr), p(xall), (rank - K) % world, K, 0.03, 0.8, p(rz), p(wrz), p(rzc), p(wrzc),
                                                                       None, p(gbuf), 1, stream))
    run("bwd_finish", lambda: lib.crossclr_backward_finish(pp, p(gbuf), p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(inv), 0.03, p(go), p(gv), p(gt), gv.stride(0), gt.stride(0), stream))
    tot = sum(stages.values())
    val = (b * world) ** 2 / (tot * 1e-3)
    base = base or val
    print(f"N={world}: " + " ".join(f"{k}={x:.3f}" for k, x in stages.items()) + f" | compute {tot:.3f} ms/step -> {val:.3e} pairs/s = {val/base:.2f}x of N=1 "
          f"(all-gather payload in: {(world-1)*plan.operand_bytes/1e6:.0f} MB" + (f"; partner gradients out: {K * 2 * plan.bpad * plan.Dpad * 4 / 1e6:.0f} MB" if K else "") + ")")
