"""The fragment-major saved backwards on the MI355X (crossclr_normalize_xf -> crossclr_backward_saved_xfp: two tiles per barrier interval,
the kernel a training step runs; crossclr_backward_saved_xf: one tile per interval; include/crossclr.h ABI 5):
the column tiles go from global memory straight into MFMA B fragments through inline-asm buffer loads whose completion is counted by
hand (s_waitcnt vmcnt).  The host emulation cannot see a wrong count -- the hardware can: every accumulator receives the same MFMA
sequence as in the LDS-staged kernel, so the two gradient buffers must agree BIT FOR BIT, launch after launch, for every DK
instantiation, with mirrored and direct tiles, with and without sample weights.  (Against the REFERENCE these kernels are pinned by
tests/test_gpu_parity.py::test_large_cases_sampled_rows, which asserts that it ran one of them.)"""
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
    for entry in (lib.crossclr_backward_saved_xfp, lib.crossclr_backward_saved_xf):
        for it in range(30 if B <= 4096 else 12):
            g_xf = torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
            nat.check(entry(pp, p(ws.xf), p(ws.stash), ws.temperature, ws.negative_w, p(ws.rz), p(ws.wrz), sw, p(g_xf), 0, stream))
            torch.cuda.synchronize()
            assert torch.equal(g_xf, g_lds), (it, (g_xf - g_lds).abs().max().item())
        # accumulate = 1 adds on top of what is there
        nat.check(entry(pp, p(ws.xf), p(ws.stash), ws.temperature, ws.negative_w, p(ws.rz), p(ws.wrz), sw, p(g_xf), 1, stream))
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


def test_the_module_self_tests_the_fragment_major_backwards_and_falls_back(monkeypatch):
    """loss._saved_backward_entry: before a process takes a fragment-major backward at a kernel instantiation it runs loss._xf_selftest
    (synthetic batches, candidate vs LDS-staged kernel, bit for bit) -- outside the training step: the step itself launches ONE saved
    backward.  Agreement -> the pair kernel from then on; a candidate that is wrong (simulated by feeding it a zeroed operand copy) -> a
    warning and the next candidate: crossclr_backward_saved_xf, then the LDS-staged kernel; the gradients never change."""
    import warnings
    lib = nat.library()
    B, D = 2304, 512
    g = torch.Generator().manual_seed(11)
    v0, t0 = torch.randn(B, D, generator=g).cuda(), torch.randn(B, D, generator=g).cuda()

    def step():
        v, t = v0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
        crossclr_amd.crossclr_loss(v, t, 0.05, 0.8, compute_mode="bf16").backward()
        return v.grad, t.grad
    # (the step itself runs inside the library -- crossclr_step_backward -- so its choice is read from the layout the library reported;
    #  the self-test drives the fine-grained entry points from Python, which is where the counters and the simulated defects sit)
    calls = {"xfp": 0, "xf": 0, "lds": 0}
    real = {"xfp": lib.crossclr_backward_saved_xfp, "xf": lib.crossclr_backward_saved_xf, "lds": lib.crossclr_backward_saved}
    count = lambda k: (lambda *a: (calls.__setitem__(k, calls[k] + 1), real[k](*a))[1])
    for k, name in (("xfp", "crossclr_backward_saved_xfp"), ("xf", "crossclr_backward_saved_xf"), ("lds", "crossclr_backward_saved")):
        monkeypatch.setattr(lib, name, count(k))
    monkeypatch.setattr(L, "_xf_verified", {})
    gv, gt = step()
    selftest_lds = calls["lds"]                       # the self-test's reference launches (one per synthetic batch)
    assert calls["xfp"] >= 2 and calls["xf"] == 0 and selftest_lds == len(L._XF_SELFTEST_ROWS) and list(L._xf_verified.values()) == [True]
    assert L._last_step_backward_kernel == 3          # the pair kernel
    before = dict(calls)
    gv2, gt2 = step()
    assert calls == before and L._last_step_backward_kernel == 3 and torch.equal(gv, gv2) and torch.equal(gt, gt2)      # verified once
    # a build whose pair kernel is wrong: the one-tile kernel takes over
    zeros = torch.zeros(nat.make_plan(B, D, 1, 0, nat.MODE_BF16).xf_bytes, dtype=torch.uint8, device="cuda")
    broken = lambda k: (lambda pp, xf, *rest: (calls.__setitem__(k, calls[k] + 1), real[k](pp, ctypes.c_void_p(zeros.data_ptr()), *rest))[1])
    monkeypatch.setattr(lib, "crossclr_backward_saved_xfp", broken("xfp"))
    monkeypatch.setattr(L, "_xf_verified", {})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        gv3, gt3 = step()
    assert any("crossclr_backward_saved_xfp disagrees" in str(x.message) for x in w)
    assert sorted(L._xf_verified.values()) == [False, True] and L._last_step_backward_kernel == 2
    assert torch.equal(gv3, gv) and torch.equal(gt3, gt)
    before = dict(calls)
    gv4, _ = step()
    assert calls == before and L._last_step_backward_kernel == 2 and torch.equal(gv4, gv)
    # both wrong: the LDS-staged kernel
    monkeypatch.setattr(lib, "crossclr_backward_saved_xf", broken("xf"))
    monkeypatch.setattr(L, "_xf_verified", {})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        gv5, gt5 = step()
    assert sum("disagrees" in str(x.message) for x in w) == 2 and list(L._xf_verified.values()) == [False, False]
    assert L._last_step_backward_kernel == 1 and torch.equal(gv5, gv) and torch.equal(gt5, gt)
    before = dict(calls)
    gv6, _ = step()
    assert calls == before and L._last_step_backward_kernel == 1 and torch.equal(gv6, gv)
