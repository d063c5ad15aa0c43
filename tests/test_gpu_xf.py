"""The fragment-major saved backward on the MI355X (crossclr_normalize_xf -> crossclr_backward_saved_xf, include/crossclr.h ABI 4):
the column tiles go from global memory straight into MFMA B fragments through inline-asm buffer loads whose completion is counted by
hand (s_waitcnt vmcnt).  The host emulation cannot see a wrong count -- the hardware can: every accumulator receives the same MFMA
sequence as in the LDS-staged kernel, so the two gradient buffers must agree BIT FOR BIT, launch after launch, for every DK
instantiation, with mirrored and direct tiles, with and without sample weights."""
import ctypes

import pytest
import torch

import crossclr_amd
from crossclr_amd import _native as nat
from crossclr_amd import loss as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,D,weighted", [(2048, 128, False), (2048, 256, True), (1536, 384, False), (8192, 512, False), (3000, 500, True),
                                          (300, 40, False), (130, 512, False), (1024, 1024, True), (1536, 768, False), (8192, 1024, False)])
def test_fragment_major_backward_equals_the_lds_staged_one_bit_for_bit(B, D, weighted):
    lib = nat.library()
    g = torch.Generator().manual_seed(B + D)
    v = torch.randn(B, D, generator=g).cuda()
    t = torch.randn(B, D, generator=g).cuda()
    ns = lw = None
    if weighted:
        keep = lambda: (torch.rand(B, generator=g) > 0.2).float().cuda()
        ns = (keep(), keep())
        lw = (torch.rand(B, generator=g).cuda() + 0.5, torch.rand(B, generator=g).cuda() + 0.5)
    _, ws = L._forward_impl(v, t, 0.05, 0.8, "bf16", None, ns, lw, save_for_backward=True)
    plan = ws.plan
    assert plan.xf_bytes == plan.operand_bytes and ws.stash is not None
    pp, p, stream = ctypes.byref(plan), L._ptr, L._stream_for(v)
    if ws.xf is None:      # (the module's policy takes the fragment-major path at some widths only; the library offers it at all of them)
        ws.xf = torch.empty(plan.xf_bytes, dtype=torch.uint8, device="cuda")
        xh = torch.empty_like(ws.xhat)
        nat.check(lib.crossclr_normalize_xf(pp, p(v), p(t), v.stride(0), t.stride(0), ws.in_dtype, p(xh), p(ws.xf), p(ws.inv_norm), p(ws.diag), stream))
        torch.cuda.synchronize()
        assert torch.equal(xh, ws.xhat)
    sw = L._sw(ws.k_rows, ws.k_rows, None)
    n = plan.gbuf_bytes // 4
    g_lds = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    nat.check(lib.crossclr_backward_saved(pp, p(ws.xhat), p(ws.stash), ws.temperature, ws.negative_w, p(ws.rz), p(ws.wrz), sw, p(g_lds), 0, stream))
    torch.cuda.synchronize()
    assert torch.isfinite(g_lds).all()
    for it in range(30 if B <= 4096 else 12):
        g_xf = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
        nat.check(lib.crossclr_backward_saved_xf(pp, p(ws.xf), p(ws.stash), ws.temperature, ws.negative_w, p(ws.rz), p(ws.wrz), sw, p(g_xf), 0,
                                                 stream))
        torch.cuda.synchronize()
        assert torch.equal(g_xf, g_lds), (it, (g_xf - g_lds).abs().max().item())
    # accumulate = 1 adds on top of what is there
    nat.check(lib.crossclr_backward_saved_xf(pp, p(ws.xf), p(ws.stash), ws.temperature, ws.negative_w, p(ws.rz), p(ws.wrz), sw, p(g_xf), 1, stream))
    torch.cuda.synchronize()
    assert torch.equal(g_xf, g_lds + g_lds)


def test_the_fragment_major_copy_is_the_packed_operand_rearranged():
    """xhat_xf[u][dt][ks][32 h + n][e] = xhat[32 u + 16 ks + 8 (e >> 2) + 4 h + (e & 3)][32 dt + n]  (include/crossclr.h)"""
    B, D = 2500, 500      # (the module's policy takes the fragment-major pair from 2048 padded rows on)
    g = torch.Generator().manual_seed(3)
    v, t = torch.randn(B, D, generator=g).cuda(), torch.randn(B, D, generator=g).cuda()
    for pre in (False, True):
        a, b = (torch.nn.functional.normalize(v, dim=1), torch.nn.functional.normalize(t, dim=1)) if pre else (v, t)
        _, ws = L._forward_impl(a, b, 0.05, 0.8, "bf16", None, None, None, save_for_backward=True, prenormalized=pre)
        plan = ws.plan
        x = ws.xhat.view(torch.bfloat16).view(2 * plan.bpad, plan.Dpad)
        xf = ws.xf.view(torch.bfloat16).view(2 * plan.bpad // 32, plan.Dpad // 32, 2, 2, 32, 8)    # [u][dt][ks][h][n][e]
        e = torch.arange(8, device="cuda")
        want = torch.empty_like(xf)
        for ks in range(2):
            for h in range(2):
                rows = 16 * ks + 8 * (e >> 2) + 4 * h + (e & 3)                                   # [e]
                blk = x.view(2 * plan.bpad // 32, 32, plan.Dpad // 32, 32)[:, rows]               # [u][e][dt][n]
                want[:, :, ks, h] = blk.permute(0, 2, 3, 1)
        assert torch.equal(xf, want)
        # and the row-major operand is what the plain entry point writes
        xh2 = torch.empty_like(ws.xhat); inv = torch.empty_like(ws.inv_norm); dg = torch.empty_like(ws.diag)
        entry = nat.library().crossclr_pack if pre else nat.library().crossclr_normalize
        nat.check(entry(ctypes.byref(plan), L._ptr(a), L._ptr(b), a.stride(0), b.stride(0), ws.in_dtype, L._ptr(xh2), L._ptr(inv), L._ptr(dg),
                        L._stream_for(a)))
        torch.cuda.synchronize()
        assert torch.equal(xh2, ws.xhat) and torch.equal(inv, ws.inv_norm) and torch.equal(dg, ws.diag)


def test_the_module_verifies_the_fragment_major_backward_once_per_process_and_falls_back(monkeypatch):
    """loss._saved_backward_kernel: the first step at a kernel instantiation runs both saved backwards and compares them bit for bit;
    agreement -> the fragment-major one from then on; a difference (simulated here by feeding it a zeroed operand copy) -> a warning, the
    LDS-staged kernel for the rest of the process, and correct gradients on that very step."""
    import warnings
    lib = nat.library()
    B, D = 2304, 512
    g = torch.Generator().manual_seed(11)
    v0, t0 = torch.randn(B, D, generator=g).cuda(), torch.randn(B, D, generator=g).cuda()

    def step():
        v, t = v0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
        crossclr_amd.crossclr_loss(v, t, 0.05, 0.8, compute_mode="bf16").backward()
        return v.grad, t.grad
    calls = {"xf": 0, "lds": 0}
    real_xf, real_lds = lib.crossclr_backward_saved_xf, lib.crossclr_backward_saved
    monkeypatch.setattr(lib, "crossclr_backward_saved_xf", lambda *a: (calls.__setitem__("xf", calls["xf"] + 1), real_xf(*a))[1])
    monkeypatch.setattr(lib, "crossclr_backward_saved", lambda *a: (calls.__setitem__("lds", calls["lds"] + 1), real_lds(*a))[1])
    monkeypatch.setattr(L, "_xf_verified", {})
    gv, gt = step()
    assert calls == {"xf": 1, "lds": 1} and list(L._xf_verified.values()) == [True]
    gv2, gt2 = step()
    assert calls == {"xf": 2, "lds": 1} and torch.equal(gv, gv2) and torch.equal(gt, gt2)
    # a build whose fragment-major kernel is wrong
    zeros = torch.zeros(nat.make_plan(B, D, 1, 0, nat.MODE_BF16).xf_bytes, dtype=torch.uint8, device="cuda")
    monkeypatch.setattr(lib, "crossclr_backward_saved_xf",
                        lambda pp, xf, *rest: (calls.__setitem__("xf", calls["xf"] + 1), real_xf(pp, ctypes.c_void_p(zeros.data_ptr()), *rest))[1])
    monkeypatch.setattr(L, "_xf_verified", {})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        gv3, gt3 = step()
    assert any("fragment-major" in str(x.message) for x in w) and list(L._xf_verified.values()) == [False]
    assert torch.equal(gv3, gv) and torch.equal(gt3, gt)
    before = dict(calls)
    gv4, _ = step()
    assert calls["xf"] == before["xf"] and calls["lds"] == before["lds"] + 1 and torch.equal(gv4, gv)
