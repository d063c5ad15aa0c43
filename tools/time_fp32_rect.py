#!/usr/bin/env python3
"""Exact-fp32 sharded run, one remote block (rank 0 of 2, driven through the C-ABI on one GPU): the generic forward that saves its fp32
exponentials + the saved backward (bwd_saved32_kernel<..., RECT>) against the recomputing pair; tau below 0.0078 = the two-pass regime
(U and Ut saved, bwd_saved32_kernel<..., RM, RECT>); mode bf16 with D > 1024: the wide bf16 plans' remote block (bf16 records of the generic
forward, D-slice backward in column parts).  usage: time_fp32_rect.py [b] [D] [tau] [fp32|bf16]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crossclr_amd
from crossclr_amd import _profile, _native as nat
from bench import make_inputs
b = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
tau = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
mode = sys.argv[4] if len(sys.argv) > 4 else "fp32"
peak = 157.3 if mode == "fp32" else 2500.0
v, t = make_inputs(2 * b, D, 3)
r = _profile.remote_block_times(v.cuda(), t.cuda(), tau, 0.8, iters=7, warmup=3, mode=nat.MODE_FP32 if mode == "fp32" else nat.MODE_BF16)
for k in ("forward_recompute_path", "forward_rect_save", "backward_recompute", "backward_rect_saved"):
    print(f"b={b} D={D} {mode} tau={tau}{' (two-pass)' if r['two_pass'] else ''}, one remote block: {k}: {r[k]:.3f} ms", flush=True)
print(f"stash {r['stash_bytes'] / 2**30:.2f} GiB; saved backward = {8.0 * b * b * D / (r['backward_rect_saved'] * 1e-3) / 1e12:.1f} TF alg "
      f"({8.0 * b * b * D / (r['backward_rect_saved'] * 1e-3) / 1e12 / peak:.1%} of the {mode} peak)")
