#!/usr/bin/env python3
"""Is the step power-limited?  Times the forward / backward kernels on random rows and on all-zero rows (same instruction
stream, no operand toggling in the MFMA data path): a large gap = the chip's power management, not the schedule, sets the time.
usage: power_probe.py [B] [D]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crossclr_amd import _profile
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
g = torch.Generator().manual_seed(1)
for name, v, t in (("randn", torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)),
                   ("zeros", torch.zeros(B, D), torch.zeros(B, D)),
                   ("randn", torch.randn(B, D, generator=g), torch.randn(B, D, generator=g))):
    st = _profile.stage_times(v.cuda(), t.cuda(), 0.03, 0.8, "bf16", iters=30, warmup=10)
    print(f"{name}: forward_save={st.get('forward_save', 0):.4f} ms backward_saved={st.get('backward_saved', 0):.4f} ms "
          f"forward={st['forward']:.4f} backward(recompute)={st['backward']:.4f}")
