#!/usr/bin/env python3
"""Where the host time of an eager step goes at a launch-bound size (B=256, D=512): wall time per call of each layer."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from crossclr_amd import _native as nat, loss as L
B, D = 256, 512
g = torch.Generator().manual_seed(1)
v = torch.randn(B, D, generator=g).cuda(); t = torch.randn(B, D, generator=g).cuda()
crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
def timeit(name, fn, n=2000):
    for _ in range(50): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name:46s} host {1e6*(t1-t0)/n:7.1f} us/call   (+ drain {1e6*(t2-t1)/n:6.1f} us/call)")
lib = nat.library(); plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16); pp = ctypes.byref(plan)
x = torch.empty(plan.operand_bytes, dtype=torch.uint8, device="cuda"); inv = torch.empty(2*plan.bpad, device="cuda"); dg = torch.empty(plan.bpad, device="cuda")
st = L._stream_for(v)
timeit("one ctypes launch (crossclr_normalize)", lambda: lib.crossclr_normalize(pp, v.data_ptr(), t.data_ptr(), D, D, nat.IN_F32, x.data_ptr(), inv.data_ptr(), dg.data_ptr(), st))
timeit("torch.empty(1 MB)", lambda: torch.empty(1 << 20, dtype=torch.uint8, device="cuda"))
timeit("_forward_impl (no save)", lambda: L._forward_impl(v, t, 0.03, 0.8, "bf16", None))
timeit("_forward_impl (save)", lambda: L._forward_impl(v, t, 0.03, 0.8, "bf16", None, save_for_backward=True))
def nograd():
    with torch.no_grad(): crit(v, t)
timeit("module forward under no_grad", nograd)
vg, tg = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
timeit("module forward with grad (graph node built)", lambda: crit(vg, tg))
def step():
    vg.grad = tg.grad = None
    crit(vg, tg).backward()
timeit("fwd + bwd", step)
lossk = crit(vg, tg)
timeit("backward only (retain_graph)", lambda: lossk.backward(retain_graph=True))
# eager step, wall clock with the GPU drained, at the launch-bound sizes
for Bx in (256, 1024, 2048):
    vb = torch.randn(Bx, D, generator=g).cuda().requires_grad_(True); tb = torch.randn(Bx, D, generator=g).cuda().requires_grad_(True)
    def stepb():
        vb.grad = tb.grad = None
        crit(vb, tb).backward()
    for _ in range(100): stepb()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(1000): stepb()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"B={Bx}: eager fwd + bwd: {1e6*(t1-t0)/1000:7.1f} us/step (wall, GPU drained)")
# the same loop with autograd's multithreading switched off (the backward then runs on the calling thread: no hand-over)
for Bx in (256, 1024, 2048):
    vb = torch.randn(Bx, D, generator=g).cuda().requires_grad_(True); tb = torch.randn(Bx, D, generator=g).cuda().requires_grad_(True)
    def stepb():
        vb.grad = tb.grad = None
        crit(vb, tb).backward()
    with torch.autograd.set_multithreading_enabled(False):
        for _ in range(100): stepb()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(1000): stepb()
        torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"B={Bx}: eager fwd + bwd, torch.autograd.set_multithreading_enabled(False): {1e6*(t1-t0)/1000:7.1f} us/step")
