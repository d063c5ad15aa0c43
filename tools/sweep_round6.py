#!/usr/bin/env python3
"""Randomised parity sweep of round 6's new device paths.  usage: sweep_round6.py [cases] [seed]
  1. crossclr_second_order (create_graph=True) against autograd through the op-for-op float64 oracle: random B <= 320, 2 <= D <= 1100 (D = 1 is degenerate: unit 'vectors' are +-1 and the reference's own double backward returns rounding noise), tau in
     [0.004, 0.2] (both soft-max regimes), w in [0, 2], random cotangents -- every padding / slice-width combination of the hvp kernels;
  2. MaxMargin_coot: the backward from the saved hinge mask against the recomputing backward, bit for bit, random B <= 3000, D <= 1100, both modes."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import crossclr_amd
from oracle import crossclr_oracle as orc
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 6)
worst = 0.0
for i in range(N):
    B = rng.choice([1, 2, 7, 33, 64, 65, 100, 127, 128, 129, 200, 256, 257, 320])
    D = rng.choice([2, 3, 16, 48, 63, 64, 65, 128, 130, 200, 256, 257, 300, 512, 513, 700, 768, 1000, 1100])
    tau = rng.choice([0.004, 0.006, 0.01, 0.03, 0.05, 0.1, 0.2])
    w = rng.choice([0.0, 0.5, 0.8, 1.0, 2.0])
    v, t = orc.make_inputs(rng.choice(["randn", "cluster"]), B, D, 100 + i)
    uv, ut = orc.make_inputs("randn", B, D, 5000 + i)

    def run(loss_fn, dev):
        vv, tt = v.to(dev).requires_grad_(True), t.to(dev).requires_grad_(True)
        gv, gt = torch.autograd.grad(loss_fn(vv, tt), (vv, tt), create_graph=True)
        s = (uv.to(dev).double() * gv.double()).sum() + (ut.to(dev).double() * gt.double()).sum()
        return [x.double().cpu() for x in torch.autograd.grad(s, (vv, tt))]
    got = run(lambda a, b: crossclr_amd.crossclr_loss(a, b, tau, w, compute_mode="fp32"), "cuda")
    want = run(lambda a, b: orc.eager_loss(a, b, tau, w), "cpu")
    # tolerance: 5e-4 of the largest entry of the reference's H u, plus the fp32 floor of the closed form -- entries of H are sums of terms of
    # size (1 / tau^2) / B that cancel where the soft-max saturates (the float64 reference then returns exact zeros)
    ref = max(want[0].abs().max().item(), want[1].abs().max().item())
    if ref != ref or ref == float("inf"):      # (e.g. D = 3, tau = 0.004, w = 2: the positive pair's probability underflows float64, -log 0 = inf)
        print(f"second-order B={B:4d} D={D:4d} tau={tau:<5} w={w}: the reference's own result is not finite -- skipped")
        continue
    floor = 3e-7 * max(uv.abs().max().item(), ut.abs().max().item()) / (B * tau * tau)
    err = max((got[0] - want[0]).abs().max().item(), (got[1] - want[1]).abs().max().item())
    ratio = err / (5e-4 * ref + floor)
    worst = max(worst, ratio)
    flag = "" if ratio <= 1.0 else "   <-- ABOVE THE BAR"
    print(f"second-order B={B:4d} D={D:4d} tau={tau:<5} w={w}: max |Hu - ref| = {err:.2e}, max |ref| = {ref:.2e}, bar {5e-4 * ref + floor:.2e}{flag}")
print(f"second-order: worst error / bar over {N} cases {worst:.2f}")
bad = 0
for i in range(N):
    B = rng.choice([1, 5, 64, 100, 128, 129, 500, 1000, 2048, 3000])
    D = rng.choice([3, 16, 64, 100, 256, 300, 512, 768, 1100])
    mode = rng.choice(["fp32", "bf16"]) if D <= 1024 else "fp32"
    margin = rng.choice([0.0, 0.05, 0.1, 0.3])
    im, s = orc.make_inputs(rng.choice(["randn", "cluster"]), B, D, 200 + i)
    im, s = torch.nn.functional.normalize(im, dim=1).cuda(), torch.nn.functional.normalize(s, dim=1).cuda()

    def mm():
        a, b = im.clone().requires_grad_(True), s.clone().requires_grad_(True)
        loss = crossclr_amd.max_margin_loss(a, b, margin, compute_mode=mode)
        saved = loss.grad_fn.sc.mask is not None
        loss.backward()
        return loss.item(), a.grad, b.grad, saved
    os.environ.pop("CROSSCLR_MAXMARGIN_SAVE", None)
    l1, a1, b1, s1 = mm()
    os.environ["CROSSCLR_MAXMARGIN_SAVE"] = "0"
    l0, a0, b0, s0 = mm()
    ok = s1 and not s0 and l1 == l0 and torch.equal(a1, a0) and torch.equal(b1, b0)
    bad += 0 if ok else 1
    print(f"max-margin   B={B:4d} D={D:4d} {mode} margin={margin}: saved-mask backward {'== recomputing backward (bits)' if ok else 'DIFFERS'}")
print(f"max-margin: {bad} of {N} cases differ")
sys.exit(1 if (bad or worst > 1.0) else 0)
