#!/usr/bin/env python3
"""Step time of the fused projection head + criterion (ProjectedCrossCLR: SURVEY.md 8(f) rank 2) against the unfused form
(nn.Linear x 2 under bf16 autocast + CrossCLR_onlyIntraModality).  usage: projection_bench.py [b] [Din] [D]
Run under `rocprofv3 --kernel-trace --stats` to see which kernels a projected step launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crossclr_amd
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
din = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
D = int(sys.argv[3]) if len(sys.argv) > 3 else 512
torch.manual_seed(0)
xv, xt = torch.randn(b, din, device="cuda"), torch.randn(b, din, device="cuda")
fused = crossclr_amd.ProjectedCrossCLR(din, din, D).cuda()
lin_v, lin_t = torch.nn.Linear(din, D).cuda(), torch.nn.Linear(din, D).cuda()
lin_v.load_state_dict(fused.video_proj.state_dict()); lin_t.load_state_dict(fused.text_proj.state_dict())
crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()

def step_fused():
    xv.grad = xt.grad = None
    fused.zero_grad(set_to_none=True)
    loss = fused(xv, xt); loss.backward(); return loss

def step_unfused():
    xv.grad = xt.grad = None
    lin_v.zero_grad(set_to_none=True); lin_t.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yv, yt = lin_v(xv), lin_t(xt)
    loss = crit(yv.float(), yt.float()); loss.backward(); return loss

def timed(fn, warm=30, n=50):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): last = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, last.item()

xgrad = os.environ.get("PROJ_XGRAD") == "1"      # the projections' inputs require a gradient too (an encoder in front): dx = g_y W is formed
only = os.environ.get("PROJ_BENCH")      # "fused" / "unfused": one path only (for a kernel trace of that path)
for xdt in (torch.float32, torch.bfloat16):
    xv, xt = xv.detach().to(xdt).requires_grad_(xgrad), xt.detach().to(xdt).requires_grad_(xgrad)
    mf, lf = timed(step_fused) if only != "unfused" else (float("nan"), float("nan"))
    mu, lu = timed(step_unfused) if only != "fused" else (float("nan"), float("nan"))
    print(f"b={b} Din={din} D={D} inputs {str(xdt).split('.')[-1]}{' (requiring grad)' if xgrad else ''}: fused projection + loss {mf:.3f} ms/step (loss {lf:.5f}); "
          f"nn.Linear (bf16 autocast) + loss {mu:.3f} ms/step (loss {lu:.5f})", flush=True)
