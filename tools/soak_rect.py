#!/usr/bin/env python3
"""Launch-to-launch bit-identity of the sharded step's kernels through the C-ABI (one GPU plays rank `world // 2`): the pairs /
antipode forwards with saved exponentials, the rectangular saved backward and its transposed (partner-gradient) form, the recomputing backward over a rank range."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crossclr_amd import _native as nat, loss as L
lib, p = nat.library(), L._ptr
f32 = dict(dtype=torch.float32, device="cuda")
bad_total = 0
for world, b, D in ((4, 512, 128), (4, 512, 256), (4, 1024, 512), (5, 512, 384), (8, 256, 256), (4, 512, 1024), (3, 512, 768), (8, 1024, 512)):
    rank = world // 2
    g = torch.Generator().manual_seed(world * 1000 + D)
    plan = nat.make_plan(b, D, world, rank, nat.MODE_BF16); pp = ctypes.byref(plan)
    n2 = 2 * plan.bpad
    xall = torch.empty(world * plan.operand_bytes, dtype=torch.uint8, device="cuda")
    inv, dg = torch.empty(n2, **f32), torch.empty(plan.bpad, **f32)
    st = None
    for r in range(world):
        v = torch.randn(b, D, generator=g).cuda(); t = torch.randn(b, D, generator=g).cuda()
        st = L._stream_for(v)
        nat.check(lib.crossclr_normalize(pp, p(v), p(t), D, D, nat.IN_F32, p(xall[r * plan.operand_bytes:]), p(inv), p(dg), st))
    xr = xall[rank * plan.operand_bytes:(rank + 1) * plan.operand_bytes]
    K = (world - 1) // 2
    rz = torch.rand(world * n2, generator=g).cuda() * 1e-3; wrz = rz * 0.8
    rzl, wrzl = rz[rank * n2:(rank + 1) * n2].contiguous(), wrz[rank * n2:(rank + 1) * n2].contiguous()
    ref = None
    bad = 0
    for it in range(40):
        part = torch.zeros(plan.fwd_ws_floats, **f32)
        colsum = torch.zeros(max(K, 1) * n2, **f32)
        stp = torch.zeros(lib.crossclr_rect_stash_bytes(pp, max(K, 1)), dtype=torch.uint8, device="cuda")
        sta = torch.zeros(lib.crossclr_rect_stash_bytes(pp, 1), dtype=torch.uint8, device="cuda")
        gbuf = torch.zeros(plan.gbuf_bytes // 4, **f32)
        gpart = torch.zeros(max(K, 1) * (plan.gbuf_bytes // 4), **f32)
        if K:
            nat.check(lib.crossclr_forward_rect_save(pp, p(xr), p(xall), (rank + 1) % world, K, 1, 0.05, 0.8, None, p(part), plan.fwd_slots, p(colsum), p(stp), st))
        opp = (rank + world // 2) % world
        if world % 2 == 0:
            nat.check(lib.crossclr_forward_rect_save(pp, p(xr), p(xall), opp, 1, 0, 0.05, 0.8, None, p(part), 2 * plan.fwd_slots, None, p(sta), st))
        if K:
            nat.check(lib.crossclr_backward_rect_saved(pp, p(xall), p(stp), (rank + 1) % world, K, 0.05, 0.8, p(rzl), p(wrzl), p(rz), p(wrz), None, p(gbuf), 0, st))
            for u in range(K):      # the partners' halves of the pair blocks, from the same stash
                nat.check(lib.crossclr_backward_rect_saved_t(pp, p(xr), p(stp), (rank + 1) % world, K, u, 0.05, 0.8, p(rzl), p(wrzl), p(rz), p(wrz), None,
                                                             p(gpart[u * (plan.gbuf_bytes // 4):]), st))
            nat.check(lib.crossclr_backward_ranks(pp, p(xr), p(xall), (rank - K) % world, K, 0.05, 0.8, p(rzl), p(wrzl), p(rz), p(wrz), None, p(gbuf), 1, st))
        if world % 2 == 0:
            nat.check(lib.crossclr_backward_rect_saved(pp, p(xall), p(sta), opp, 1, 0.05, 0.8, p(rzl), p(wrzl), p(rz), p(wrz), None, p(gbuf), 1, st))
        torch.cuda.synchronize()
        # (slots of the launch groups that ran; the workspace beyond them is never read)
        cur = (part[plan.fwd_slots * n2: 3 * plan.fwd_slots * n2].clone(), colsum, stp, sta, gbuf, gpart)
        if ref is None: ref = cur
        else:
            for name, a, c in zip(("part", "colsum", "stash_pairs", "stash_antipode", "gbuf", "gpartner"), cur, ref):
                if not torch.equal(a, c):
                    bad += 1
                    if bad <= 3: print(f"  world={world} b={b} D={D} iteration {it}: {name} differs in {(a != c).sum().item()} elements")
    bad_total += bad
    print(f"world={world} b={b} D={D}: {'OK' if bad == 0 else str(bad) + ' MISMATCHES'}")
sys.exit(1 if bad_total else 0)
