#!/usr/bin/env python3
"""Step time of the shapes outside the register-resident kernels' range (review item "slow shapes"): wide embeddings (D > 1024, bf16),
bf16 plans in the two-pass (small temperature) regime, against the D = 512 / 1024 steps of the same box.  usage: slow_shapes_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crossclr_amd
from bench import make_inputs
CASES = [(8192, 512, 0.03, "bf16"), (8192, 1024, 0.03, "bf16"), (8192, 1536, 0.03, "bf16"), (8192, 2048, 0.03, "bf16"),
         (8192, 512, 0.005, "bf16"), (8192, 512, 0.005, "fp32"), (8192, 1100, 0.03, "bf16"), (8192, 1536, 0.005, "bf16")]
for B, D, tau, mode in CASES:
    v, t = make_inputs(B, D, 1234)
    v, t = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    crit = crossclr_amd.CrossCLR_onlyIntraModality(tau, 0.8, compute_mode=mode).cuda()
    for _ in range(4):
        v.grad = t.grad = None; l = crit(v, t); l.backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 8
    for _ in range(n):
        v.grad = t.grad = None; l = crit(v, t); l.backward()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    peak = 2500.0 if mode == "bf16" else 157.3
    print(f"B={B} D={D} tau={tau} {mode}: {ms:.3f} ms/step = {14.0*B*B*D/(ms*1e-3)/1e12:.0f} TF alg ({14.0*B*B*D/(ms*1e-3)/1e12/peak:.1%} of peak) loss {l.item():.6f}", flush=True)
