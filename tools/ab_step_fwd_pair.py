#!/usr/bin/env python3
"""Same-process A/B of the whole training step with the two symmetric forwards (CROSSCLR_FWD_PAIR is read per launch): blocks of N steps,
alternating, HIP events around each block -- box-to-box spread (+-5 %) and clock drift cancel.  usage: ab_step_fwd_pair.py [B] [D] [N] [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
R = int(sys.argv[4]) if len(sys.argv) > 4 else 6
g = torch.Generator().manual_seed(1234)
v = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
t = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()


def step():
    v.grad = t.grad = None
    loss = crit(v, t)
    loss.backward()
    return loss


def block(n):
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        step()
    z.record()
    torch.cuda.synchronize()
    return a.elapsed_time(z) / n


for _ in range(300):
    step()
res = {"pipe (round 4)": [], "pair (round 5)": []}
for r in range(R):
    for name, val in (("pipe (round 4)", "0"), ("pair (round 5)", "1")):
        os.environ["CROSSCLR_FWD_PAIR"] = val
        block(20)
        res[name].append(block(N))
for name, xs in res.items():
    xs2 = sorted(xs)
    print(f"{name:16s} ms/step per block: " + " ".join(f"{x:.4f}" for x in xs) + f" | median {xs2[len(xs2) // 2]:.4f} min {xs2[0]:.4f}")
a, b = sorted(res["pipe (round 4)"]), sorted(res["pair (round 5)"])
print(f"B={B} D={D}: pair / pipe = {b[len(b) // 2] / a[len(a) // 2]:.4f} (medians), step saves {(a[len(a) // 2] - b[len(b) // 2]) * 1e3:.1f} us")
