#!/usr/bin/env python3
"""Step time when the boundary is handed HOST buffers: H2D copy of both [B, D] fp32 inputs + fwd+bwd + D2H of both grads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from oracle import crossclr_oracle as orc
B, D = 8192, 512
crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
v, t = orc.make_inputs("randn", B, D, 1234)
for pinned in (False, True):
    hv, ht = (v.pin_memory(), t.pin_memory()) if pinned else (v, t)
    gv_h = torch.empty_like(v).pin_memory() if pinned else torch.empty_like(v)
    gt_h = torch.empty_like(t).pin_memory() if pinned else torch.empty_like(t)
    def step():
        dv = hv.to("cuda", non_blocking=True).requires_grad_(True)
        dt = ht.to("cuda", non_blocking=True).requires_grad_(True)
        crit(dv, dt).backward()
        gv_h.copy_(dv.grad, non_blocking=True)
        gt_h.copy_(dt.grad, non_blocking=True)
    for _ in range(3): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    torch.cuda.synchronize()
    dt_ = (time.perf_counter() - t0) / 20
    print(f"{'pinned' if pinned else 'pageable'} host buffers: {dt_*1e3:.3f} ms/step -> {B*B/dt_:.3e} pairs/s (67 MB over PCIe per step)")
