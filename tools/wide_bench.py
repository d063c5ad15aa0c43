import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import crossclr_amd
from crossclr_amd import loss as L
B = 8192
for D in (4096, 5120, 6144, 8192):
    g = torch.Generator().manual_seed(1)
    v = torch.randn(B, D, generator=g).cuda().requires_grad_(True); t = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
    def step():
        v.grad = t.grad = None
        l = crit(v, t); l.backward(); return l
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): l = step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"B={B} D={D}: {ms:.2f} ms/step (backward kernel {L._last_step_backward_kernel}, saved {L._last_step_saved}), loss {l.item():.6f}, "
          f"{14.0 * B * B * D / (ms * 1e-3) / 1e12:.0f} TF algorithmic", flush=True)
