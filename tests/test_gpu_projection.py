"""SURVEY.md 8(f) rank 2 on the MI355X: the caller-side entry points against the float64 oracle / the reference's goldens.
  * prenormalized=True (crossclr_pack + crossclr_backward_finish_p): unit rows from `F.normalize` upstream under autograd;
  * ProjectedCrossCLR / projected_crossclr_loss (crossclr_project_pack): projection + L2-norm + pack in ONE launch -- the
    normalize kernel must not run on that path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import crossclr_amd
from conftest import golden_arrays, golden_index, golden_inputs
from crossclr_amd import _native as nat
from oracle import crossclr_oracle as orc

pytestmark = pytest.mark.gpu
IDX = golden_index()


@pytest.mark.parametrize("name,mode", [("g3_b256_d512_s2", "fp32"), ("g3_b256_d512_s2", "bf16"), ("g1_b64_d256_s0", "fp32"),
                                       ("g7_b4096_d512_s1234", "bf16")])
def test_prenormalized_entry_matches_the_reference_goldens(name, mode):
    """Unit rows skip crossclr_normalize; with F.normalize upstream (autograd) loss and INPUT gradients must equal the reference's
    (goldens from /root/reference/trainer/loss.py, tests/golden/make_golden.py) -- not merely the plain path of this library."""
    m = IDX[name]
    v, t = golden_inputs(m)
    vd, td = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    loss = crossclr_amd.crossclr_loss(F.normalize(vd, dim=1), F.normalize(td, dim=1), m["temperature"], m["negative_weight"],
                                      compute_mode=mode, prenormalized=True)
    loss.backward()
    torch.cuda.synchronize()
    assert loss.dtype == torch.float64
    assert abs(loss.item() - m["loss"]) <= (2e-5 * max(1.0, abs(m["loss"])) if mode == "fp32" else 1e-3)
    arr = golden_arrays(name)
    scale = max(m["grad_v_absmax"], m["grad_t_absmax"])
    tol = (2e-4 if mode == "fp32" else 1e-2) * scale
    if "grad_v" in arr:
        assert np.abs(vd.grad.cpu().numpy() - arr["grad_v"]).max() <= tol and np.abs(td.grad.cpu().numpy() - arr["grad_t"]).max() <= tol
    else:
        rows = arr["rows"]
        assert np.abs(vd.grad[rows].cpu().numpy() - arr["grad_v_rows"]).max() <= tol
        assert np.abs(td.grad[rows].cpu().numpy() - arr["grad_t_rows"]).max() <= tol


def _reference(xv, xt, wv, bv, wt, bt, tau, w):
    args = [a.double().clone().requires_grad_(True) if a is not None else None for a in (xv, xt, wv, bv, wt, bt)]
    loss = orc.eager_loss(F.linear(args[0], args[2], args[3]), F.linear(args[1], args[4], args[5]), tau, w)
    loss.backward()
    return loss.item(), [a.grad if a is not None else None for a in args]


@pytest.mark.parametrize("b,din_v,din_t,D,bias", [(2048, 768, 512, 512, True), (1000, 300, 200, 256, False), (512, 1024, 1024, 128, True),
                                                  (1024, 512, 300, 768, True), (2048, 1024, 768, 1024, False),     # 32 rows per block
                                                  (4096, 512, 512, 512, True)])    # the fragment-major pair backward behind the projection
def test_fused_projection_matches_float64_autograd(b, din_v, din_t, D, bias, monkeypatch):
    g = torch.Generator().manual_seed(b + D)
    xv, xt = torch.randn(b, din_v, generator=g), torch.randn(b, din_t, generator=g)
    wv, wt = torch.randn(D, din_v, generator=g) / din_v ** 0.5, torch.randn(D, din_t, generator=g) / din_t ** 0.5
    bv = 0.1 * torch.randn(D, generator=g) if bias else None
    bt = 0.1 * torch.randn(D, generator=g) if bias else None
    ref_loss, ref = _reference(xv, xt, wv, bv, wt, bt, 0.03, 0.8)
    leaves = [a.cuda().requires_grad_(True) if a is not None else None for a in (xv, xt, wv, bv, wt, bt)]
    # the separate normalisation pass must not be launched on this path
    lib = nat.library()
    def boom(*a):
        raise AssertionError("crossclr_normalize was launched on the fused-projection path")
    monkeypatch.setattr(lib, "crossclr_normalize", boom, raising=False)
    monkeypatch.setattr(lib, "crossclr_pack", boom, raising=False)
    # ... nor the two-step normalise-backward: the finish kernel writes g_y itself (prenormalized = 2)
    monkeypatch.setattr(lib, "crossclr_project_backward_prep", boom, raising=False)
    loss = crossclr_amd.projected_crossclr_loss(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], leaves[5], 0.03, 0.8)
    loss.backward()
    torch.cuda.synchronize()
    assert loss.dtype == torch.float64 and loss.dim() == 0
    assert abs(loss.item() - ref_loss) <= 2e-3 * max(1.0, abs(ref_loss))      # bf16 products in the projection AND the similarities
    for got, want, name in zip(leaves, ref, ("x_video", "x_text", "w_video", "b_video", "w_text", "b_text")):
        if want is None:
            continue
        scale = want.abs().max().item()
        err = (got.grad.double().cpu() - want).abs().max().item()
        assert err <= 2e-2 * scale, (name, err, scale)


def test_projected_module_is_deterministic_and_graph_safe():
    torch.manual_seed(1)
    crit = crossclr_amd.ProjectedCrossCLR(256, 384, 512).cuda()
    xv, xt = torch.randn(2048, 256).cuda(), torch.randn(2048, 384).cuda()
    ref = None
    for _ in range(5):
        crit.zero_grad()
        loss = crit(xv, xt)
        loss.backward()
        cur = (loss.detach().clone(), crit.video_proj.weight.grad.clone(), crit.text_proj.bias.grad.clone())
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(cur, ref))


@pytest.mark.parametrize("name", ["g6_aligned_b2048_d512", "g6_aligned_b256_d128"])
def test_bf16_gradients_in_the_aligned_regime(name):
    """Near-zero loss (aligned pairs): the float64 reference gradient is tiny and tells nothing about a bf16 run; the yardstick is
    the bf16-operand model (oracle.bf16_operand_model_loss_and_grads: rounded operands, straight-through gradient)."""
    m = IDX[name]
    v, t = golden_inputs(m)
    ml, mgv, mgt = orc.bf16_operand_model_loss_and_grads(v, t, m["temperature"], m["negative_weight"])
    vd, td = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
    loss = crossclr_amd.crossclr_loss(vd, td, m["temperature"], m["negative_weight"], compute_mode="bf16")
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(ml)) <= 1e-4 * max(1.0, abs(float(ml))) + 1e-6
    scale = max(mgv.abs().max().item(), mgt.abs().max().item())
    # Here the gradient is the difference of terms of size g0 = 1 / (2 B tau) that cancel to ~1e-4 g0 (the positive pair's weight is
    # 1 - O(1e-6)); the backward's weights are bf16, so what can be promised is an ABSOLUTE accuracy relative to g0 -- 1e-4 g0,
    # five times what is measured -- on top of the usual relative bar
    g0 = 1.0 / (2.0 * m["B"] * m["temperature"])
    tol = 2e-2 * scale + 1e-4 * g0
    assert (vd.grad.double().cpu() - mgv).abs().max().item() <= tol
    assert (td.grad.double().cpu() - mgt).abs().max().item() <= tol
    assert max(vd.grad.abs().max().item(), td.grad.abs().max().item()) <= scale + 1e-4 * g0      # and it stays tiny
