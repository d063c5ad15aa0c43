#!/usr/bin/env python3
"""Quick check of a (minimal, -DCROSSCLR_DSL_MINIMAL) tuning build of the pair kernel: bit-identity with the LDS-staged kernel at a few
Dpad = 512 shapes (unweighted), then HIP-event times of the three saved backwards at B = 8192.  usage: check_xfp.py [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from crossclr_amd import _native as nat, loss as L
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
lib = nat.library()
bad_total = 0
for B, D in [(130, 512), (384, 500), (640, 512), (1000, 480), (2048, 512), (4096, 512), (8192, 512)]:
    g = torch.Generator().manual_seed(B * 7 + D)
    v, t = torch.randn(B, D, generator=g).cuda(), torch.randn(B, D, generator=g).cuda()
    plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16)
    pp, p = ctypes.byref(plan), L._ptr
    stream = torch.cuda.current_stream().cuda_stream
    n2 = 2 * plan.bpad
    f32 = dict(dtype=torch.float32, device="cuda")
    xhat = torch.empty(plan.operand_bytes, dtype=torch.uint8, device="cuda")
    xf = torch.empty(plan.xf_bytes, dtype=torch.uint8, device="cuda")
    inv_norm, diag = torch.empty(n2, **f32), torch.empty(plan.bpad, **f32)
    logz, rz, wrz = torch.empty(n2, **f32), torch.empty(n2, **f32), torch.empty(n2, **f32)
    part = torch.empty(plan.fwd_ws_floats, **f32)
    loss_sum = torch.empty(max(2, plan.loss_ws_doubles), dtype=torch.float64, device="cuda")
    stash = torch.empty(plan.stash_bytes, dtype=torch.uint8, device="cuda")
    nat.check(lib.crossclr_normalize_xf(pp, p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(xhat), p(xf), p(inv_norm), p(diag), stream))
    nat.check(lib.crossclr_forward_save(pp, p(xhat), 0.03, 0.8, None, p(part), 0, p(stash), stream))
    nat.check(lib.crossclr_forward_finish_w(pp, p(part), plan.fwd_slots, p(diag), 0.03, 0.8, None, p(logz), p(rz), p(wrz), p(loss_sum), stream))
    want = torch.empty(plan.gbuf_bytes // 4, **f32)
    nat.check(lib.crossclr_backward_saved(pp, p(xhat), p(stash), 0.03, 0.8, p(rz), p(wrz), None, p(want), 0, stream))
    bad = 0
    for i in range(reps):
        got = torch.full_like(want, float("nan"))
        nat.check(lib.crossclr_backward_saved_xfp(pp, p(xf), p(stash), 0.03, 0.8, p(rz), p(wrz), None, p(got), 0, stream))
        if not torch.equal(got, want):
            bad += 1
            if bad <= 2:
                d = (got != want) | (got.isnan())
                print(f"  B={B} D={D} launch {i}: {d.sum().item()} of {d.numel()} elements differ; first at {d.nonzero()[0].item()}; nan {got.isnan().sum().item()}")
    bad_total += bad
    print(f"B={B} D={D}: {reps} launches, {'bit-identical' if bad == 0 else str(bad) + ' MISMATCHES'}", flush=True)
    if B == 8192:
        def timed(fn, n=20):
            for _ in range(3): nat.check(fn())
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
            for a, z in ev:
                a.record(); nat.check(fn()); z.record()
            torch.cuda.synchronize()
            ms = sorted(a.elapsed_time(z) for a, z in ev)
            return ms[len(ms) // 2], sum(ms) / len(ms)
        for rep in range(3):
            for name, fn in (("xfp", lambda: lib.crossclr_backward_saved_xfp(pp, p(xf), p(stash), 0.03, 0.8, p(rz), p(wrz), None, p(got), 0, stream)),
                             ("xf ", lambda: lib.crossclr_backward_saved_xf(pp, p(xf), p(stash), 0.03, 0.8, p(rz), p(wrz), None, p(got), 0, stream)),
                             ("lds", lambda: lib.crossclr_backward_saved(pp, p(xhat), p(stash), 0.03, 0.8, p(rz), p(wrz), None, p(got), 0, stream))):
                med, avg = timed(fn)
                print(f"  {name}: median {med:.4f} ms  avg {avg:.4f} ms  ({8.0 * B * B * D / med / 1e9:.0f} TF)")
print("check_xfp:", "OK" if bad_total == 0 else f"{bad_total} MISMATCHES")
