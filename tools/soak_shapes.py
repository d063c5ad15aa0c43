#!/usr/bin/env python3
"""Repeat-launch consistency over many shapes: loss and gradients must be finite and bit-identical on every step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
shapes = [(2048, 256), (2048, 128), (2048, 384), (1000, 256), (512, 256), (4096, 256), (2048, 64), (300, 96), (2048, 768), (1024, 1024)]
if len(sys.argv) > 2: shapes = [(int(sys.argv[1]), int(sys.argv[2]))]
for B, D in shapes:
    g = torch.Generator().manual_seed(B + D)
    v = torch.randn(B, D, generator=g).cuda().requires_grad_(True); t = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.05, 0.8, compute_mode="bf16").cuda()
    ref, bad = None, 0
    for i in range(400):
        v.grad = t.grad = None
        loss = crit(v, t); loss.backward()
        if i % 10 == 0 or True:
            cur = (loss.item(), v.grad.double().sum().item(), t.grad.abs().double().sum().item())
            if ref is None: ref = cur
            if cur != ref:
                bad += 1
                if bad <= 3: print(f"  B={B} D={D} step {i}: {cur} != {ref}")
    print(f"B={B} D={D}: {'OK' if bad == 0 else str(bad) + ' MISMATCHES'} ref={ref}")
