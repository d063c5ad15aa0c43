#!/usr/bin/env python3
"""The criterion under torch.cuda.make_graphed_callables (forward and backward as two HIP graphs sharing a memory pool) against the eager
step: nothing behind the C-ABI synchronises or allocates, so the module is graph-safe as it is.  usage: graphed_step.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crossclr_amd
from bench import make_inputs
for B, D in ((256, 512), (1024, 512), (2048, 512), (4096, 512), (8192, 512)):
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
    v, t = make_inputs(B, D, 1)
    v, t = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    def eager():
        v.grad = t.grad = None
        loss = crit(v, t); loss.backward(); return loss
    for _ in range(30): eager()
    torch.cuda.synchronize()
    n = 300 if B <= 4096 else 100
    t0 = time.perf_counter()
    for _ in range(n): le = eager()
    torch.cuda.synchronize()
    te = (time.perf_counter() - t0) / n
    ge = (v.grad.clone(), t.grad.clone())
    sv, st = v.detach().clone().requires_grad_(True), t.detach().clone().requires_grad_(True)
    gcrit = torch.cuda.make_graphed_callables(crit, (sv, st), allow_unused_input=True)      # (logit_scale is a parameter the forward never reads)
    def graphed():
        v.grad = t.grad = None
        loss = gcrit(v, t); loss.backward(); return loss
    for _ in range(30): graphed()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): lg = graphed()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / n
    same = lg.item() == le.item() and torch.equal(v.grad, ge[0]) and torch.equal(t.grad, ge[1])
    print(f"B={B} D={D}: eager {te*1e6:.1f} us/step, make_graphed_callables {tg*1e6:.1f} us/step, bit-identical: {same}", flush=True)
