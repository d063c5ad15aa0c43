#!/usr/bin/env python3
"""Soak of the fragment-major saved backward: at every shape, `reps` launches of crossclr_backward_saved_xf must all be
bit-identical to ONE launch of the LDS-staged crossclr_backward_saved over the same stash -- a wrong hand-counted vmcnt would show as an
occasional difference in whole fragments.  usage: soak_xf.py [reps] [entry = crossclr_backward_saved_xfp | crossclr_backward_saved_xf] [B,D ...]
(entry defaults to the pair kernel; shapes default to the list below; a stash of 4 GiB or more is skipped for the pair kernel)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CROSSCLR_XF_WIDTHS", "128,256,384,512,768,1024,1152,1536,2048,2560,3072,4096")
import torch, crossclr_amd
from crossclr_amd import _native as nat, loss as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
entry_name = sys.argv[2] if len(sys.argv) > 2 else "crossclr_backward_saved_xfp"
lib = nat.library()
entry = getattr(lib, entry_name)
shapes = [(8192, 512), (8192, 448), (4096, 512), (2048, 512), (1000, 500), (384, 512), (130, 512), (8192, 384), (3000, 300), (8192, 256), (2048, 200),
          (8192, 128), (4096, 64), (300, 40), (16384, 512), (8192, 1024), (2048, 768), (1000, 900),
          # wide plans (pair kernel in 3 ... 8 column parts; crossclr_backward_saved_xfp only)
          (4096, 1100), (8192, 1536), (2048, 2048), (1000, 2300), (640, 3000), (512, 4096)]
if len(sys.argv) > 3:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[3:]]
bad_total = 0
for B, D in shapes:
    g = torch.Generator().manual_seed(B * 7 + D)
    v, t = torch.randn(B, D, generator=g).cuda(), torch.randn(B, D, generator=g).cuda()
    for weighted in (False, True):
        ns = lw = None
        if weighted:
            keep = lambda: (torch.rand(B, generator=g) > 0.2).float().cuda()
            ns, lw = (keep(), keep()), (torch.rand(B, generator=g).cuda() + 0.5, torch.rand(B, generator=g).cuda() + 0.5)
        _, ws = L._forward_impl(v, t, 0.05, 0.8, "bf16", None, ns, lw, save_for_backward=True)
        plan = ws.plan
        assert ws.xf is not None and ws.stash is not None
        if plan.Dpad > 1024 and not entry_name.endswith("xfp"):
            print(f"B={B} D={D}: wide plan, pair kernel only, skipped"); continue
        if entry_name.endswith("xfp") and plan.stash_bytes >= (1 << 32):
            print(f"B={B} D={D}: stash of {plan.stash_bytes} bytes, skipped"); continue
        pp, p, stream = ctypes.byref(plan), L._ptr, L._stream_for(v)
        sw = L._sw(ws.k_rows, ws.k_rows, None)
        n = plan.gbuf_bytes // 4
        g_lds = torch.empty(n, dtype=torch.float32, device="cuda")
        nat.check(lib.crossclr_backward_saved(pp, p(ws.xhat), p(ws.stash), ws.temperature, ws.negative_w, p(ws.rz), p(ws.wrz), sw, p(g_lds), 0, stream))
        g_xf = torch.empty(n, dtype=torch.float32, device="cuda")
        bad = 0
        r = max(20, reps // 10) if B >= 8192 else reps
        for i in range(r):
            g_xf.fill_(float("nan"))
            nat.check(entry(pp, p(ws.xf), p(ws.stash), ws.temperature, ws.negative_w, p(ws.rz), p(ws.wrz), sw, p(g_xf), 0, stream))
            if not torch.equal(g_xf, g_lds):
                bad += 1
                if bad <= 2: print(f"  B={B} D={D} w={weighted} launch {i}: {(g_xf != g_lds).sum().item()} elements differ, max {(g_xf - g_lds).abs().max().item():.3e}")
        bad_total += bad
        print(f"B={B} D={D} weighted={weighted}: {r} launches, {'all bit-identical to the LDS-staged kernel' if bad == 0 else str(bad) + ' MISMATCHES'}")
print(f"soak_xf ({entry_name}):", "OK" if bad_total == 0 else f"{bad_total} MISMATCHES")
sys.exit(1 if bad_total else 0)
