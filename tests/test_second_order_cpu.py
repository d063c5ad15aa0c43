"""CPU tests (host emulation of the same kernel sources): the double backward of the criterion -- crossclr_second_order (include/crossclr.h,
csrc/crossclr_kernels_hvp.h) behind `create_graph=True` -- against golden vectors generated from the REFERENCE's own double backward
(tests/golden/make_golden_second_order.py: trainer/loss.py:79-114 is a chain of eager ops, autograd differentiates its backward again)
and, for what the reference does not have (per-sample weights, unit rows given as such), against autograd through the float64 oracle."""
import json
import os

import numpy as np
import pytest
import torch

import crossclr_amd
from crossclr_amd import _native as nat
from oracle import crossclr_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {m["name"]: m for m in json.load(open(os.path.join(GOLDEN, "so_index.json")))["cases"]}


@pytest.fixture(scope="module", autouse=True)
def emulated_library():
    from emu import build_emu
    nat.use_library_for_testing(build_emu.build())
    yield
    nat.use_library_for_testing(None)


def second_order_through_the_module(m, dev="cpu", mode="fp32", **kw):
    v, t = orc.make_inputs(m["kind"], m["B"], m["D"], m["seed"])
    uv, ut = orc.make_inputs("randn", m["B"], m["D"], m["cotangent_seed"])
    vv, tt = v.to(dev).requires_grad_(True), t.to(dev).requires_grad_(True)
    uv, ut = uv.to(dev), ut.to(dev)
    loss = crossclr_amd.crossclr_loss(vv, tt, m["temperature"], m["negative_weight"], compute_mode=mode, **kw)
    gv, gt = torch.autograd.grad(loss, (vv, tt), create_graph=True)
    assert gv.requires_grad and gt.requires_grad
    s = (uv.double() * gv.double()).sum() + (ut.double() * gt.double()).sum()
    hv, ht = torch.autograd.grad(s, (vv, tt), retain_graph=True)
    pen = (gv.double() ** 2).sum() + 0.5 * (gt.double() ** 2).sum()
    pv, pt = torch.autograd.grad(pen, (vv, tt))
    return {k: x.detach().cpu() for k, x in dict(loss=loss, gv=gv, gt=gt, s=s, hv=hv, ht=ht, pv=pv, pt=pt).items()}


# the emulated kernels run at ~1e5 tile products per second: the small and the two-pass cases here, every case on the GPU (tests/test_gpu_second_order.py)
@pytest.mark.parametrize("name", ["so_b8_d16_s1", "so_b64_d48_s5", "so_w0_tau01_b16_d32_s3", "so_tau0005_b32_d64_s9", "so_ragged_b100_d200_s6"])
def test_double_backward_against_the_reference_goldens(name):
    m = CASES[name]
    want = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    got = second_order_through_the_module(m)
    assert abs(got["loss"].item() - m["loss"]) <= 2e-5 * max(1.0, abs(m["loss"]))
    assert abs(got["s"].item() - m["s"]) <= 1e-4 * max(abs(m["s"]), np.abs(want["gv"]).max() * m["D"] ** 0.5)
    for key, tol in (("gv", 2e-5), ("gt", 2e-5), ("hv", 2e-4), ("ht", 2e-4), ("pv", 2e-4), ("pt", 2e-4)):
        scale = max(np.abs(want[key[0] + "v"]).max(), np.abs(want[key[0] + "t"]).max())
        assert np.abs(got[key].double().numpy() - want[key]).max() <= tol * scale, key


def _weighted_oracle(v, t, tau, w, k, om, prenormalized):
    """float64 closed form with per-sample weights (the reference has none): Z_p = sum_q e^{A_pq} + sum_{q != p} k_q e^{w S_pq} + k_p e^0,
    L = sum_p omega_p (log Z_p - A_pp) / 2B -- differentiable, autograd does the rest."""
    F = torch.nn.functional
    b = v.shape[0]
    vn, tn = (v, t) if prenormalized else (F.normalize(v, dim=1), F.normalize(t, dim=1))
    vn, tn = vn.double(), tn.double()
    off = 1.0 - torch.eye(b, dtype=torch.float64)
    inter = vn @ tn.t() / tau

    def side(a, x, kk, oo):
        intra = (x @ x.t()) / tau * off * w
        ea = torch.exp(intra) * (kk.double()[None, :] if kk is not None else 1.0)
        nll = torch.log(torch.exp(a).sum(1) + ea.sum(1)) - a.diagonal()
        return (nll * (oo.double() if oo is not None else 1.0)).sum()
    return (side(inter, vn, k[0] if k else None, om[0] if om else None) + side(inter.t(), tn, k[1] if k else None, om[1] if om else None)) / (2.0 * b)


@pytest.mark.parametrize("weighted,prenormalized,tau", [(True, False, 0.05), (False, True, 0.05), (True, False, 0.005)])
def test_double_backward_with_sample_weights_and_unit_rows(weighted, prenormalized, tau):
    B, D = 24, 20
    v, t = orc.make_inputs("randn", B, D, 4)
    uv, ut = orc.make_inputs("randn", B, D, 1004)
    if prenormalized:
        v, t = torch.nn.functional.normalize(v, dim=1), torch.nn.functional.normalize(t, dim=1)
    g = torch.Generator().manual_seed(3)
    k = (torch.rand(B, generator=g) + 0.25, (torch.rand(B, generator=g) > 0.2).float()) if weighted else None
    om = (torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) * 2.0) if weighted else None

    def run(loss_fn):
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        gv, gt = torch.autograd.grad(loss_fn(vv, tt), (vv, tt), create_graph=True)
        s = (uv.double() * gv.double()).sum() + (ut.double() * gt.double()).sum()
        hv, ht = torch.autograd.grad(s, (vv, tt))
        return gv.detach().double(), gt.detach().double(), hv.double(), ht.double()
    got = run(lambda a, b: crossclr_amd.crossclr_loss(a, b, tau, 0.8, compute_mode="fp32", negative_scale=k, loss_weight=om,
                                                      prenormalized=prenormalized))
    want = run(lambda a, b: _weighted_oracle(a, b, tau, 0.8, k, om, prenormalized))
    for gg, ww, tol in zip(got, want, (2e-5, 2e-5, 3e-4, 3e-4)):
        assert (gg - ww).abs().max().item() <= tol * max(want[0].abs().max().item(), ww.abs().max().item())


def test_grad_out_and_single_cotangent_and_third_order():
    """d<u, g>/d(grad_out) = <u, dL/d(rows)>; a cotangent for one of the two gradients only; scaling the loss scales H u; third order raises."""
    v, t = orc.make_inputs("randn", 16, 24, 2)
    u, _ = orc.make_inputs("randn", 16, 24, 1002)
    vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    scale = torch.tensor(2.5, dtype=torch.float64, requires_grad=True)
    loss = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="fp32") * scale
    gv, = torch.autograd.grad(loss, vv, create_graph=True)
    s = (u.double() * gv.double()).sum()
    hv, ht, hs = torch.autograd.grad(s, (vv, tt, scale))
    rv, rt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    rs = torch.tensor(2.5, dtype=torch.float64, requires_grad=True)
    rg, = torch.autograd.grad(orc.eager_loss(rv, rt, 0.05, 0.8) * rs, rv, create_graph=True)
    wv, wt, ws = torch.autograd.grad((u.double() * rg.double()).sum(), (rv, rt, rs))
    assert (hv.double() - wv.double()).abs().max().item() <= 2e-4 * wv.abs().max().item()
    assert (ht.double() - wt.double()).abs().max().item() <= 2e-4 * wt.abs().max().item()
    assert abs(hs.item() - ws.item()) <= 1e-5 * max(1.0, abs(ws.item()))
    vv2, tt2 = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    g2, = torch.autograd.grad(crossclr_amd.crossclr_loss(vv2, tt2, 0.05, 0.8, compute_mode="fp32"), vv2, create_graph=True)
    with pytest.raises(RuntimeError, match="third-order"):                  # (a graph through the double backward: refused, loudly)
        torch.autograd.grad((g2 ** 2).sum(), vv2, create_graph=True)


def test_second_order_entry_point_refuses_what_it_does_not_cover():
    import ctypes
    lib = nat.library()
    assert lib.crossclr_second_order_workspace_bytes(ctypes.byref(nat.make_plan(64, 32, 1, 0, nat.MODE_BF16))) == 0      # exact-fp32 plans only
    assert b"FP32" in lib.crossclr_last_error()
    assert lib.crossclr_second_order_workspace_bytes(ctypes.byref(nat.make_plan(64, 32, 2, 0, nat.MODE_FP32))) == 0      # single device
    assert lib.crossclr_second_order_workspace_bytes(ctypes.byref(nat.make_plan(64, 32, 1, 0, nat.MODE_FP32))) > 0


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 3e-4), (torch.bfloat16, 3e-2), (torch.float16, 1e-2)])
def test_double_backward_in_every_input_dtype(dtype, tol):
    """Cotangents arrive and Hessian-vector products leave in the INPUT dtype (like the gradients); the arithmetic in between is the exact-fp32
    closed form whatever the inputs are -- compared with autograd through the float64 oracle on the same (rounded) inputs."""
    v, t = orc.make_inputs("randn", 24, 40, 6, dtype)
    uv, ut = orc.make_inputs("randn", 24, 40, 1006, dtype)

    def run(loss_fn, cast):
        vv, tt = cast(v).requires_grad_(True), cast(t).requires_grad_(True)
        gv, gt = torch.autograd.grad(loss_fn(vv, tt), (vv, tt), create_graph=True)
        s = (cast(uv).double() * gv.double()).sum() + (cast(ut).double() * gt.double()).sum()
        hv, ht = torch.autograd.grad(s, (vv, tt))
        return hv, ht
    got = run(lambda a, b: crossclr_amd.crossclr_loss(a, b, 0.05, 0.8, compute_mode="fp32"), lambda x: x.clone())
    want = run(lambda a, b: orc.eager_loss(a, b, 0.05, 0.8), lambda x: x.double().clone())
    assert got[0].dtype == dtype and got[1].dtype == dtype
    scale = max(want[0].abs().max().item(), want[1].abs().max().item())
    assert (got[0].double() - want[0]).abs().max().item() <= tol * scale and (got[1].double() - want[1]).abs().max().item() <= tol * scale


@pytest.mark.parametrize("B,D", [(24, 300), (40, 520)])
def test_double_backward_over_rows_wider_than_one_slice(B, D):
    """Dpad = 512 / 768: the product pass owns two 256-column output slices per block (the tile's S and T evaluated once) resp. three blocks of one."""
    v, t = orc.make_inputs("randn", B, D, 8)
    uv, ut = orc.make_inputs("randn", B, D, 1008)

    def run(loss_fn):
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        gv, gt = torch.autograd.grad(loss_fn(vv, tt), (vv, tt), create_graph=True)
        s = (uv.double() * gv.double()).sum() + (ut.double() * gt.double()).sum()
        return [x.double() for x in torch.autograd.grad(s, (vv, tt))]
    got = run(lambda a, b: crossclr_amd.crossclr_loss(a, b, 0.05, 0.8, compute_mode="fp32"))
    want = run(lambda a, b: orc.eager_loss(a, b, 0.05, 0.8))
    scale = max(want[0].abs().max().item(), want[1].abs().max().item())
    assert (got[0] - want[0]).abs().max().item() <= 3e-4 * scale and (got[1] - want[1]).abs().max().item() <= 3e-4 * scale
