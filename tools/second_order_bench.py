#!/usr/bin/env python3
"""Timing of the double backward (crossclr_second_order: create_graph=True through the criterion).  usage: second_order_bench.py [B] [D]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crossclr_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = torch.Generator().manual_seed(1)
v = torch.randn(B, D, generator=g).cuda(); t = torch.randn(B, D, generator=g).cuda()
for mode in ("fp32", "bf16"):
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode=mode).cuda()
    def step():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        gv, gt = torch.autograd.grad(crit(vv, tt), (vv, tt), create_graph=True)
        pen = (gv.double() ** 2).sum() + (gt.double() ** 2).sum()
        return torch.autograd.grad(pen, (vv, tt))
    for _ in range(2): step()
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    # executed flops of the two hvp passes + the fp32 first-order pieces: (3 + 3 * Dpad/128 + 2) GEMM units of 2 (2B)^2 D each + forward/backward
    print(f"B={B} D={D} first-order mode {mode}: gradient-penalty step (forward + backward + double backward) {ms:.2f} ms, peak memory "
          f"{torch.cuda.max_memory_allocated() / 2 ** 20:.0f} MiB (the eager form: {3 * 8 * B * 2 * B * 4 / 2 ** 30:.1f} GiB of float64 [B, 2B] tensors per level)")
