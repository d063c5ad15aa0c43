#!/usr/bin/env python3
"""Step time in the two-pass (small temperature) regime, exact-fp32 products.  usage: small_tau_bench.py  (saved exponentials);
CROSSCLR_DISABLE_SAVE=1 small_tau_bench.py  (recomputing backward; the library reads its tuning variables once per process)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crossclr_amd
from bench import make_inputs
for B in (4096, 8192):
    v, t = make_inputs(B, 512, 1234)
    v, t = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.005, 0.8, compute_mode="fp32").cuda()
    for _ in range(3):
        v.grad = t.grad = None; l = crit(v, t); l.backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        v.grad = t.grad = None; l = crit(v, t); l.backward()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} D=512 tau=0.005 fp32 {'recompute' if os.environ.get('CROSSCLR_DISABLE_SAVE') else 'saved'}: "
          f"{e0.elapsed_time(e1)/5:.3f} ms/step loss {l.item():.6f} |gv| {v.grad.norm().item():.6e}")
