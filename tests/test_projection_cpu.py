"""Producer-side fusion on the emulated kernels (tiny shapes): projection + L2-norm + pack in one launch (crossclr_project_pack),
the normalise-backward in front of the projection's backward (crossclr_project_backward_prep).  Checker: float64 autograd through
`F.linear` and the oracle's op-for-op restatement of the reference loss (trainer/loss.py:76-114)."""
import pytest
import torch
import torch.nn.functional as F

import crossclr_amd
from crossclr_amd import _native as nat
from oracle import crossclr_oracle as orc


@pytest.fixture(scope="module", autouse=True)
def emulated_library():
    from emu import build_emu
    nat.use_library_for_testing(build_emu.build())
    yield
    nat.use_library_for_testing(None)


def reference(xv, xt, wv, bv, wt, bt, tau, w):
    args = [a.double().clone().requires_grad_(True) if a is not None else None for a in (xv, xt, wv, bv, wt, bt)]
    loss = orc.eager_loss(F.linear(args[0], args[2], args[3]), F.linear(args[1], args[4], args[5]), tau, w)
    loss.backward()
    return loss.item(), [a.grad if a is not None else None for a in args]


@pytest.mark.parametrize("b,din_v,din_t,D,bias", [(40, 24, 40, 32, True), (70, 64, 100, 48, False), (130, 30, 30, 200, True),
                                                  (40, 64, 24, 600, True)])      # Dpad = 768: 32 rows per block
def test_fused_projection_matches_float64_autograd(b, din_v, din_t, D, bias):
    g = torch.Generator().manual_seed(b + D)
    xv, xt = torch.randn(b, din_v, generator=g), torch.randn(b, din_t, generator=g)
    wv, wt = torch.randn(D, din_v, generator=g) / din_v ** 0.5, torch.randn(D, din_t, generator=g) / din_t ** 0.5
    bv = 0.1 * torch.randn(D, generator=g) if bias else None
    bt = 0.1 * torch.randn(D, generator=g) if bias else None
    ref_loss, ref = reference(xv, xt, wv, bv, wt, bt, 0.1, 0.8)
    leaves = [a.clone().requires_grad_(True) if a is not None else None for a in (xv, xt, wv, bv, wt, bt)]
    loss = crossclr_amd.projected_crossclr_loss(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], leaves[5], 0.1, 0.8)
    assert loss.dtype == torch.float64 and loss.dim() == 0
    loss.backward()
    assert abs(loss.item() - ref_loss) <= 5e-3 * max(1.0, abs(ref_loss))        # bf16 products in projection and similarities
    for got, want, name in zip(leaves, ref, ("x_video", "x_text", "w_video", "b_video", "w_text", "b_text")):
        if want is None:
            continue
        scale = want.abs().max().item()
        err = (got.grad.double() - want).abs().max().item()
        assert err <= 3e-2 * scale, (name, err, scale)


def test_module_owns_two_linear_layers_and_trains():
    torch.manual_seed(0)
    crit = crossclr_amd.ProjectedCrossCLR(20, 28, 32, temperature=0.1, negative_weight=0.8)
    assert {n for n, _ in crit.named_parameters()} == {"video_proj.weight", "video_proj.bias", "text_proj.weight", "text_proj.bias"}
    xv, xt = torch.randn(48, 20), torch.randn(48, 28)
    opt = torch.optim.SGD(crit.parameters(), lr=0.5)
    first = None
    for _ in range(4):
        opt.zero_grad()
        loss = crit(xv, xt)
        loss.backward()
        opt.step()
        first = first if first is not None else loss.item()
    assert loss.item() < first


def test_packed_operand_equals_normalize_of_the_projection():
    """crossclr_project_pack against crossclr_normalize fed the fp32 projection: same inv_norm / diagonal cosines to bf16-product accuracy."""
    import ctypes
    lib = nat.library()
    b, din, D = 64, 40, 64
    g = torch.Generator().manual_seed(3)
    xv, xt = torch.randn(b, din, generator=g), torch.randn(b, din, generator=g)
    wv, wt = torch.randn(D, din, generator=g) / din ** 0.5, torch.randn(D, din, generator=g) / din ** 0.5
    plan = nat.make_plan(b, D, 1, 0, nat.MODE_BF16)
    from crossclr_amd.projection import _weights_bf16
    wvb, ldw = _weights_bf16(wv)
    wtb, _ = _weights_bf16(wt)
    xhat = torch.empty(plan.operand_bytes, dtype=torch.uint8)
    inv, diag = torch.empty(2 * plan.bpad), torch.empty(plan.bpad)
    nat.check(lib.crossclr_project_pack(ctypes.byref(plan), xv.data_ptr(), xt.data_ptr(), din, din, din, din, nat.IN_F32, wvb.data_ptr(), wtb.data_ptr(),
                                        ldw, ldw, 0, 0, xhat.data_ptr(), inv.data_ptr(), diag.data_ptr(), 0))
    yv, yt = xv.double() @ wv.double().t(), xt.double() @ wt.double().t()
    want_inv = torch.cat([1 / yv.norm(dim=1), 1 / yt.norm(dim=1)])
    got_inv = torch.cat([inv[:b], inv[plan.bpad:plan.bpad + b]]).double()
    assert ((got_inv - want_inv).abs() / want_inv).max().item() < 2e-2
    want_diag = (F.normalize(yv, dim=1) * F.normalize(yt, dim=1)).sum(1)
    assert (diag[:b].double() - want_diag).abs().max().item() < 2e-2
    packed = xhat.view(torch.bfloat16).view(2, plan.bpad, plan.Dpad)
    assert (packed[0, :b, :D].double() - F.normalize(yv, dim=1)).abs().max().item() < 2e-2
    assert float(packed[:, b:, :].abs().max()) == 0.0 and float(packed[:, :, D:].abs().max()) == 0.0
    # the same launch with FRAGMENT-MAJOR weights (crossclr_project_pack_wf: what the module passes): the same bits
    wvf, _ = _weights_bf16(wv, plan.Dpad)
    wtf, _ = _weights_bf16(wt, plan.Dpad)
    assert wvf.numel() == plan.Dpad * ldw
    xhat2 = torch.empty_like(xhat)
    inv2, diag2 = torch.empty_like(inv), torch.empty_like(diag)
    nat.check(lib.crossclr_project_pack_wf(ctypes.byref(plan), xv.data_ptr(), xt.data_ptr(), din, din, din, din, nat.IN_F32, wvf.data_ptr(), wtf.data_ptr(),
                                           ldw, ldw, 0, 0, xhat2.data_ptr(), inv2.data_ptr(), diag2.data_ptr(), 0))
    assert torch.equal(xhat, xhat2) and torch.equal(inv[:b], inv2[:b]) and torch.equal(diag[:b], diag2[:b])


@pytest.mark.parametrize("b,D,din_v,din_t,dtype", [(70, 48, 40, 100, torch.float32), (200, 130, 64, 24, torch.bfloat16)])
def test_weight_gradient_kernel_equals_the_matrix_product(b, D, din_v, din_t, dtype):
    """crossclr_project_dw (split-K MFMA kernel with transposing LDS reads + reduce): dW = g_y^T x and db = column sums of g_y for both
    modalities, against float64 products of the same bf16-rounded operands."""
    import ctypes
    lib = nat.library()
    g = torch.Generator().manual_seed(b + D)
    gyv, gyt = (torch.randn(b, D, generator=g) * 0.1).bfloat16(), (torch.randn(b, D, generator=g) * 0.1).bfloat16()
    xv, xt = torch.randn(b, din_v, generator=g).to(dtype), torch.randn(b, din_t, generator=g).to(dtype)
    ws = torch.empty(lib.crossclr_project_dw_ws_floats(b, D, din_v, din_t))
    dwv, dwt = torch.full((D, din_v), float("nan")), torch.full((D, din_t), float("nan"))
    dbv, dbt = torch.full((D,), float("nan")), torch.full((D,), float("nan"))
    in_dtype = nat.IN_F32 if dtype == torch.float32 else nat.IN_BF16
    nat.check(lib.crossclr_project_dw(b, D, gyv.data_ptr(), gyt.data_ptr(), D, xv.data_ptr(), xt.data_ptr(), din_v, din_t, din_v, din_t, in_dtype,
                                      ws.data_ptr(), dwv.data_ptr(), dwt.data_ptr(), din_v, din_t, dbv.data_ptr(), dbt.data_ptr(), 0))
    for gy, x, dw, db in ((gyv, xv, dwv, dbv), (gyt, xt, dwt, dbt)):
        want = gy.double().t() @ x.bfloat16().double()
        assert (dw.double() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
        assert (db.double() - gy.double().sum(0)).abs().max().item() <= 1e-4


def test_weight_casts_track_data_updates_and_fresh_modules():
    """Round-4 review: a cache keyed on id(weight) / _version returned stale bf16 weights after `.data` updates (fused optimisers, EMA)
    and for a new module whose parameter reused a freed one's id and address.  The casts are now redone on every call."""
    from crossclr_amd.projection import _weights_bf16
    for it in range(20):
        lin = torch.nn.Linear(128, 128)
        want = lin.weight.detach().to(torch.bfloat16)
        got, ldw = _weights_bf16(lin.weight)
        assert ldw == 128 and torch.equal(got, want), it
        frag, _ = _weights_bf16(lin.weight, 128)
        back = frag.view(4, 8, 2, 32, 8).permute(0, 3, 1, 2, 4).reshape(128, 128)
        assert torch.equal(back, want), it
        del lin
    w = torch.nn.Parameter(torch.randn(64, 96))
    a, _ = _weights_bf16(w)
    w.data.mul_(2.0)                       # no _version bump
    b, _ = _weights_bf16(w)
    assert torch.equal(b[:, :96], (w.detach()).to(torch.bfloat16)) and not torch.equal(a, b)
