#!/usr/bin/env python3
"""Soak: many fwd+bwd steps on fixed inputs -- loss and gradients must be bit-identical every step (deterministic
kernels: no atomics), allocator footprint must stay flat."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from oracle import crossclr_oracle as orc
for B, D, mode, steps in ((2048, 512, "bf16", 1500), (8192, 512, "bf16", 300), (777, 300, "fp32", 300)):
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode=mode).cuda()
    v, t = orc.make_inputs("randn", B, D, 5)
    v, t = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    ref = None
    for i in range(steps):
        v.grad = t.grad = None
        loss = crit(v, t)
        loss.backward()
        if i % 50 == 0 or i == steps - 1:
            cur = (loss.item(), v.grad.double().sum().item(), t.grad.abs().double().sum().item())
            if ref is None:
                ref, mem0 = cur, torch.cuda.memory_reserved()
            assert cur == ref, (i, cur, ref)
    print(f"B={B} D={D} {mode}: {steps} steps bit-identical, reserved {mem0/2**20:.0f} -> {torch.cuda.memory_reserved()/2**20:.0f} MiB")
