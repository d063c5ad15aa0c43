#!/usr/bin/env python3
"""Eager vs HIP-graph replay latency of one fwd+bwd step at small / mid batch sizes (launch-bound regime)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from oracle import crossclr_oracle as orc
for B, D in ((256, 512), (1024, 512), (2048, 512), (4096, 512)):
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
    v, t = orc.make_inputs("randn", B, D, 1)
    v, t = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    def step():
        v.grad = t.grad = None
        crit(v, t).backward()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 200
    v.grad = t.grad = None
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        crit(v, t).backward()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 200
    print(f"B={B} D={D}: eager {eager*1e6:.1f} us/step, graph replay {rep*1e6:.1f} us/step")
