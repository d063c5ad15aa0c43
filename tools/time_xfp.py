#!/usr/bin/env python3
"""HIP-event time of crossclr_backward_saved_xfp (and _xf beside it) at B = 8192, D = 512 for the library CROSSCLR_HIP_LIBRARY names
(ablation variants produce wrong results: timing only).  usage: time_xfp.py [label]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from crossclr_amd import _native as nat, loss as L
label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("CROSSCLR_HIP_LIBRARY", "main")
B, D = 8192, 512
lib = nat.library()
g = torch.Generator().manual_seed(5)
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
got = torch.empty(plan.gbuf_bytes // 4, **f32)
nat.check(lib.crossclr_normalize_xf(pp, p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(xhat), p(xf), p(inv_norm), p(diag), stream))
nat.check(lib.crossclr_forward_save(pp, p(xhat), 0.03, 0.8, None, p(part), 0, p(stash), stream))
nat.check(lib.crossclr_forward_finish_w(pp, p(part), plan.fwd_slots, p(diag), 0.03, 0.8, None, p(logz), p(rz), p(wrz), p(loss_sum), stream))
def timed(fn, n=30):
    for _ in range(10): nat.check(fn())
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, z in ev:
        a.record(); nat.check(fn()); z.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(z) for a, z in ev)
    return ms[len(ms) // 2]
xfp = lambda: lib.crossclr_backward_saved_xfp(pp, p(xf), p(stash), 0.03, 0.8, p(rz), p(wrz), None, p(got), 0, stream)
xf1 = lambda: lib.crossclr_backward_saved_xf(pp, p(xf), p(stash), 0.03, 0.8, p(rz), p(wrz), None, p(got), 0, stream)
timed(xf1)
if label == "clocks":       # engine clock and package power while the pair kernel runs back to back (rocm-smi from a second process)
    import subprocess, time
    for _ in range(2000): nat.check(xfp())
    samples = []
    for _ in range(3):
        for _ in range(3000): nat.check(xfp())
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        samples.append(" | ".join(l.strip() for l in o.splitlines() if "sclk" in l or "Power" in l))
    torch.cuda.synchronize()
    print("# under load: " + " ## ".join(samples), flush=True)
    sys.exit(0)
r = [(timed(xfp), timed(xf1)) for _ in range(3)]
print(f"{label:28s} xfp " + " ".join(f"{a:.4f}" for a, _ in r) + "   xf " + " ".join(f"{b:.4f}" for _, b in r), flush=True)
