#!/usr/bin/env python3
"""cProfile of the host side of one eager fwd+bwd step at a launch-bound size."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from oracle import crossclr_oracle as orc
crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
v, t = orc.make_inputs("randn", 256, 512, 1)
v, t = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
def step():
    v.grad = t.grad = None
    crit(v, t).backward()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(22)
