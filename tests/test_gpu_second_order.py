"""GPU tests: the double backward of the criterion on the MI355X -- crossclr_second_order (include/crossclr.h, csrc/crossclr_kernels_hvp.h)
behind `create_graph=True` -- against EVERY golden case generated from the reference's own double backward
(tests/golden/make_golden_second_order.py), with the first-order step in exact-fp32 and in bf16 mode, and at a batch that spans several
column slices against autograd through the op-for-op float64 oracle."""
import os

import numpy as np
import pytest
import torch

import crossclr_amd
from oracle import crossclr_oracle as orc
from test_second_order_cpu import CASES, GOLDEN, second_order_through_the_module

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
def test_double_backward_against_the_reference_goldens_on_the_device(name):
    m = CASES[name]
    want = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    got = second_order_through_the_module(m, dev="cuda")
    assert abs(got["loss"].item() - m["loss"]) <= 2e-5 * max(1.0, abs(m["loss"]))
    for key, tol in (("gv", 2e-5), ("gt", 2e-5), ("hv", 2e-4), ("ht", 2e-4), ("pv", 2e-4), ("pt", 2e-4)):
        scale = max(np.abs(want[key[0] + "v"]).max(), np.abs(want[key[0] + "t"]).max())
        assert np.abs(got[key].double().numpy() - want[key]).max() <= tol * scale, key


def test_second_order_terms_behind_a_bf16_step():
    """The first-order step in bf16 mode (the headline arithmetic), the second-order terms in exact fp32: the Hessian-vector product meets
    the reference golden at the fp32 bar, the first gradient at the bf16 bar."""
    name = "so_b64_d256_s0"
    m = CASES[name]
    want = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    got = second_order_through_the_module(m, dev="cuda", mode="bf16")
    gscale = max(np.abs(want["gv"]).max(), np.abs(want["gt"]).max())
    assert np.abs(got["gv"].double().numpy() - want["gv"]).max() <= 1e-2 * gscale
    hscale = max(np.abs(want["hv"]).max(), np.abs(want["ht"]).max())
    assert np.abs(got["hv"].double().numpy() - want["hv"]).max() <= 2e-4 * hscale
    assert np.abs(got["ht"].double().numpy() - want["ht"]).max() <= 2e-4 * hscale


@pytest.mark.parametrize("B,D,tau", [(700, 160, 0.05), (520, 96, 0.004)])
def test_double_backward_over_several_column_slices(B, D, tau):
    v, t = orc.make_inputs("randn", B, D, 11)
    uv, ut = orc.make_inputs("randn", B, D, 1011)

    def run(loss_fn, dev):
        vv, tt = v.to(dev).requires_grad_(True), t.to(dev).requires_grad_(True)
        gv, gt = torch.autograd.grad(loss_fn(vv, tt), (vv, tt), create_graph=True)
        s = (uv.to(dev).double() * gv.double()).sum() + (ut.to(dev).double() * gt.double()).sum()
        hv, ht = torch.autograd.grad(s, (vv, tt))
        return hv.double().cpu(), ht.double().cpu()
    got = run(lambda a, b: crossclr_amd.crossclr_loss(a, b, tau, 0.8, compute_mode="fp32"), "cuda")
    want = run(lambda a, b: orc.eager_loss(a, b, tau, 0.8), "cpu")
    scale = max(want[0].abs().max().item(), want[1].abs().max().item())
    assert (got[0] - want[0]).abs().max().item() <= 3e-4 * scale and (got[1] - want[1]).abs().max().item() <= 3e-4 * scale


def test_gradient_penalty_training_step_at_the_headline_shape():
    """What create_graph is for: loss + lambda * ||dL/dv||^2 differentiated at B = 8192, D = 512 without a B x B tensor (the eager form keeps
    12.5 GB of float64 [B, 2B] tensors per autograd level).  No oracle finishes this size in seconds, so the size-independent properties:
    finite, bit-reproducible, bounded memory, and the Hessian it applies is SYMMETRIC (<u1, H u2> = <u2, H u1> for random u1, u2 -- a wrong
    term in the closed form breaks this) and linear (H (a u) = a H u)."""
    B, D = 8192, 512
    v, t = orc.make_inputs("randn", B, D, 1234)
    u1 = [x.cuda() for x in orc.make_inputs("randn", B, D, 1)]
    u2 = [x.cuda() for x in orc.make_inputs("randn", B, D, 2)]
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="fp32").cuda()

    def hessian_times(us):
        vv, tt = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
        gv, gt = torch.autograd.grad(crit(vv, tt), (vv, tt), create_graph=True)
        out = []
        for uv, ut in us:
            s = (uv.double() * gv.double()).sum() + (ut.double() * gt.double()).sum()
            out.append(torch.autograd.grad(s, (vv, tt), retain_graph=True))
        pen = (gv.double() ** 2).sum() + (gt.double() ** 2).sum()
        return out, torch.autograd.grad(pen, (vv, tt)), (gv.detach(), gt.detach())
    torch.cuda.reset_peak_memory_stats()
    (h1, h2, h3), pen_grad, g = hessian_times([u1, u2, [2.5 * u1[0], 2.5 * u1[1]]])
    assert torch.cuda.max_memory_allocated() < 3 * 2 ** 30
    for x in (*h1, *h2, *pen_grad):
        assert torch.isfinite(x).all()
    dot = lambda a, b: ((a[0].double() * b[0].double()).sum() + (a[1].double() * b[1].double()).sum()).item()
    nrm = lambda a: dot(a, a) ** 0.5
    assert abs(dot(u1, h2) - dot(u2, h1)) <= 1e-4 * nrm(u1) * nrm(h2)                              # symmetry
    assert max((h3[0] - 2.5 * h1[0]).abs().max().item(), (h3[1] - 2.5 * h1[1]).abs().max().item()) <= 1e-5 * max(h1[0].abs().max().item(), h1[1].abs().max().item()) * 2.5
    # the penalty's gradient is 2 H g: the same operator applied to the gradient itself
    (hg,), _, _ = hessian_times([list(g)])
    scale = max(pen_grad[0].abs().max().item(), pen_grad[1].abs().max().item())
    assert (pen_grad[0] - 2.0 * hg[0]).abs().max().item() <= 1e-4 * scale and (pen_grad[1] - 2.0 * hg[1]).abs().max().item() <= 1e-4 * scale
    (h1b, _, _), pen_grad2, _ = hessian_times([u1, u2, u1])
    assert torch.equal(h1[0], h1b[0]) and torch.equal(h1[1], h1b[1]) and torch.equal(pen_grad[0], pen_grad2[0])      # bit-reproducible
