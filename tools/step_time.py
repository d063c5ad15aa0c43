#!/usr/bin/env python3
"""ms per training step (module forward + backward, HIP events around blocks of N steps) of the library CROSSCLR_HIP_LIBRARY names -- for
A/Bs of two builds on one box, processes interleaved: tools/ab_rowkernels.sh.  usage: step_time.py [label] [B] [D] [N] [rounds] [weighted 0|1]
(weighted: crossclr_amd.CrossCLR with input-space features -- influential-sample pruning / weighting, BASELINE config 5's criterion)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
label = sys.argv[1] if len(sys.argv) > 1 else "lib"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
D = int(sys.argv[3]) if len(sys.argv) > 3 else 512
N = int(sys.argv[4]) if len(sys.argv) > 4 else 200
R = int(sys.argv[5]) if len(sys.argv) > 5 else 5
weighted = len(sys.argv) > 6 and sys.argv[6] == "1"
g = torch.Generator().manual_seed(1234)
v = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
t = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
crit = (crossclr_amd.CrossCLR(0.03, negative_weight=0.8, compute_mode="bf16") if weighted
        else crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16")).cuda()
xin = (torch.randn(B, 64, generator=g).cuda(), torch.randn(B, 64, generator=g).cuda())


def block(n):
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        v.grad = t.grad = None
        (crit(v, t, *xin) if weighted else crit(v, t)).backward()
    z.record()
    torch.cuda.synchronize()
    return a.elapsed_time(z) / n


block(300)
xs = [block(N) for _ in range(R)]
s = sorted(xs)
print(f"{label:10s} B={B} D={D} ms/step per block: " + " ".join(f"{x:.4f}" for x in xs) + f" | median {s[len(s) // 2]:.4f} min {s[0]:.4f}", flush=True)
