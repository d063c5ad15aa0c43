"""CPU tests (host emulation): the two step calls of the C-ABI (crossclr_step_plan / _forward / _backward, include/crossclr.h ABI 6 / 7) against
the stage entry points they are composed of -- same bits -- and the library's kernel-selection policy as crossclr_step_plan reports it."""
import ctypes

import pytest
import torch

import crossclr_amd
from crossclr_amd import _native as nat
from crossclr_amd import loss as L
from oracle import crossclr_oracle as orc


@pytest.fixture(scope="module", autouse=True)
def emulated_library():
    from emu import build_emu
    nat.use_library_for_testing(build_emu.build())
    yield
    nat.use_library_for_testing(None)


def _stages(v, t, tau, w, mode):
    """normalize -> forward -> forward_finish through the fine-grained entry points (separate loss-reduce launch)."""
    lib = nat.library()
    b, D = v.shape
    plan = nat.make_plan(b, D, 1, 0, mode)
    pp, p = ctypes.byref(plan), L._ptr
    f32 = dict(dtype=torch.float32)
    xhat = torch.empty(plan.operand_bytes, dtype=torch.uint8)
    inv_norm, diag = torch.empty(2 * plan.bpad, **f32), torch.empty(plan.bpad, **f32)
    part = torch.empty(plan.fwd_ws_floats, **f32)
    logz, rz, wrz = (torch.empty(2 * plan.bpad, **f32) for _ in range(3))
    loss_sum = torch.empty(max(2, plan.loss_ws_doubles), dtype=torch.float64)
    nat.check(lib.crossclr_normalize(pp, p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(xhat), p(inv_norm), p(diag), 0))
    nat.check(lib.crossclr_forward(pp, p(xhat), p(xhat), 1, 0, -1, tau, w, p(part), 0, 0))
    nat.check(lib.crossclr_forward_finish(pp, p(part), plan.fwd_slots, p(diag), tau, w, p(logz), p(rz), p(wrz), p(loss_sum), 0))
    return loss_sum[1].item(), logz


@pytest.mark.parametrize("B,D,mode", [(40, 24, "fp32"), (150, 32, "bf16"), (256, 16, "bf16"), (300, 40, "fp32")])
def test_step_forward_equals_the_stage_entry_points_bit_for_bit(B, D, mode):
    v, t = orc.make_inputs("randn", B, D, 3)
    want, logz = _stages(v, t, 0.05, 0.8, nat.MODE_FP32 if mode == "fp32" else nat.MODE_BF16)
    with torch.no_grad():
        got = crossclr_amd.crossclr_loss(v, t, 0.05, 0.8, compute_mode=mode)
    assert got.item() == want      # (the finish kernel's last block sums the block partials exactly like fwd_finish_reduce_kernel)
    _, ws = L._forward_impl(v, t, 0.05, 0.8, mode, None, save_for_backward=True)
    assert torch.equal(ws.logz, logz) and ws.step is not None and ws.step[2] is not None


def test_step_plan_reports_the_policy(monkeypatch):
    lib = nat.library()
    lay = nat.StepLayout()

    def plan_of(b, D, mode, tau=0.05, flags=0, nbytes=0):
        plan = nat.make_plan(b, D, 1, 0, mode)
        nat.check(lib.crossclr_step_plan(ctypes.byref(plan), tau, 0.8, flags, nbytes, ctypes.byref(lay)))
        return plan
    plan_of(2048, 512, nat.MODE_BF16)                       # the headline path at its smallest batch: pair kernel on the fragment-major copy
    assert (lay.saved, lay.two_pass, lay.backward_kernel) == (1, 0, 3) and lay.xf != nat.STEP_NONE and lay.stash != nat.STEP_NONE
    full = lay.total_bytes
    plan_of(2048, 512, nat.MODE_BF16, flags=nat.STEP_NO_XFP)
    assert lay.backward_kernel == 2
    plan_of(2048, 512, nat.MODE_BF16, flags=nat.STEP_NO_XFP | nat.STEP_NO_XF)
    assert lay.backward_kernel == 1 and lay.saved == 1
    plan_of(1024, 512, nat.MODE_BF16)                       # below the row floor: the plain pair (LDS-staged saved backward)
    assert (lay.saved, lay.backward_kernel) == (1, 1) and lay.xf == nat.STEP_NONE
    plan_of(2048, 256, nat.MODE_BF16)
    assert lay.backward_kernel == 1
    plan_of(2048, 512, nat.MODE_BF16, flags=nat.STEP_NO_SAVE)
    assert (lay.saved, lay.backward_kernel) == (0, 0) and lay.stash == nat.STEP_NONE and lay.total_bytes < full
    small = lay.total_bytes
    plan_of(2048, 512, nat.MODE_BF16, nbytes=full - 1)      # a workspace one byte short of the saving layout: the recomputing one
    assert lay.saved == 0 and lay.total_bytes == small
    plan = nat.make_plan(2048, 512, 1, 0, nat.MODE_BF16)
    assert lib.crossclr_step_plan(ctypes.byref(plan), 0.05, 0.8, 0, small - 1, ctypes.byref(lay)) == -4      # CROSSCLR_E_WORKSPACE
    plan_of(2048, 512, nat.MODE_BF16, flags=nat.STEP_FORWARD_ONLY)
    assert lay.saved == 0 and lay.backward_scratch_bytes == 0
    plan_of(2048, 512, nat.MODE_FP32, tau=0.004)            # two-pass regime, exact fp32: saves U and Ut
    assert (lay.two_pass, lay.saved) == (1, 1) and lay.shift != nat.STEP_NONE
    monkeypatch.setenv("CROSSCLR_MAX_STASH_GB", "0.001")
    plan_of(2048, 512, nat.MODE_BF16)
    assert lay.saved == 0
    monkeypatch.delenv("CROSSCLR_MAX_STASH_GB")
    monkeypatch.setenv("CROSSCLR_XF_WIDTHS", "256")
    plan_of(256, 200, nat.MODE_BF16)
    assert lay.backward_kernel == 3
    plan = nat.make_plan(64, 32, 2, 0, nat.MODE_BF16)
    assert lib.crossclr_step_plan(ctypes.byref(plan), 0.05, 0.8, 0, 0, ctypes.byref(lay)) == -1              # single device only


@pytest.mark.parametrize("B,D,mode,tau", [(150, 32, "bf16", 0.05), (40, 24, "fp32", 0.05), (64, 32, "fp32", 0.004)])
def test_eager_gradient_product_is_the_same_step(B, D, mode, tau, monkeypatch):
    """CROSSCLR_STEP_EAGER (the module's default with a backward to follow): the forward call also enqueues the gradient product, backward()
    runs the finish kernel alone -- same kernels, same bits; a second backward through the same graph re-runs the finish only."""
    v, t = orc.make_inputs("randn", B, D, 5)

    def step(twice=False):
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, tau, 0.8, compute_mode=mode)
        loss.backward(retain_graph=twice)
        if twice:
            g1 = vv.grad.clone()
            vv.grad = None
            (3.0 * loss).backward()
            assert torch.allclose(vv.grad, 3.0 * g1, rtol=1e-6, atol=0)
            vv.grad = g1
        return loss.item(), vv.grad, tt.grad
    le, gve, gte = step()
    lay = nat.StepLayout()
    _, ws = L._forward_impl(v, t, tau, 0.8, mode, None, save_for_backward=True)
    assert ws.step[0].flags & nat.STEP_EAGER and ws.step[0].backward_scratch_bytes == 0
    step(twice=True)
    monkeypatch.setenv("CROSSCLR_EAGER_BACKWARD", "0")
    _, ws = L._forward_impl(v, t, tau, 0.8, mode, None, save_for_backward=True)
    assert not (ws.step[0].flags & nat.STEP_EAGER) and ws.step[0].backward_scratch_bytes == ws.plan.gbuf_bytes
    ll, gvl, gtl = step()
    assert le == ll and torch.equal(gve, gvl) and torch.equal(gte, gtl)


def test_layout_regions_persistent_is_what_the_backward_reads():
    """ABI 7: the workspace is a persistent region (what crossclr_step_backward reads) and a transient one (the rest: most of the bytes)."""
    lib = nat.library()
    lay = nat.StepLayout()
    plan = nat.make_plan(2048, 512, 1, 0, nat.MODE_BF16)
    n2 = 2 * plan.bpad

    def planned(flags, tau=0.05):
        nat.check(lib.crossclr_step_plan(ctypes.byref(plan), tau, 0.8, flags, 0, ctypes.byref(lay)))
        assert lay.total_bytes == lay.persistent_bytes + lay.transient_bytes and lay.persistent_bytes % 256 == 0
        assert lay.flags == flags and lay.temperature == pytest.approx(tau) and lay.check != 0
        return {n: getattr(lay, n) for n in ("xhat", "inv_norm", "diag", "logz", "rz", "wrz", "part", "shift", "xf", "stash", "gbuf", "ticket")}
    o = planned(nat.STEP_EAGER)           # backward = the finish kernel alone: 1 / ||x|| and the gradient slices, nothing else
    keep = {n for n, off in o.items() if off != nat.STEP_NONE and off < lay.persistent_bytes}
    assert keep == {"inv_norm", "gbuf"} and lay.persistent_bytes == 4 * n2 + plan.gbuf_bytes and lay.backward_scratch_bytes == 0
    assert lay.transient_bytes > plan.stash_bytes + plan.xf_bytes      # saved exponentials + fragment-major copy: released after the forward call
    o = planned(0)                        # the backward forms the product: from `transient` (saved), or recomputed from the persistent region alone
    keep = {n for n, off in o.items() if off != nat.STEP_NONE and off < lay.persistent_bytes}
    assert keep == {"xhat", "inv_norm", "rz", "wrz"} and lay.backward_scratch_bytes == plan.gbuf_bytes
    o = planned(0, tau=0.004)
    keep = {n for n, off in o.items() if off != nat.STEP_NONE and off < lay.persistent_bytes}
    assert keep == {"xhat", "inv_norm", "rz", "wrz", "shift"}
    planned(nat.STEP_FORWARD_ONLY)
    assert lay.persistent_bytes == 0 and lay.saved == 0
    # a fragment-major copy is only laid out when a kernel that reads it may run (it used to be written and never read)
    planned(nat.STEP_NO_XFP | nat.STEP_NO_XF)
    assert lay.backward_kernel == 1 and lay.xf == nat.STEP_NONE and lay.xf_bytes == 0


def test_step_calls_refuse_a_layout_they_did_not_get_from_step_plan():
    lib = nat.library()
    v, t = orc.make_inputs("randn", 40, 24, 3)
    plan = nat.make_plan(40, 24, 1, 0, nat.MODE_FP32)
    lay = nat.StepLayout()
    nat.check(lib.crossclr_step_plan(ctypes.byref(plan), 0.05, 0.8, 0, 0, ctypes.byref(lay)))
    per, tra = torch.empty(lay.persistent_bytes, dtype=torch.uint8), torch.empty(lay.transient_bytes, dtype=torch.uint8)
    loss_ws = torch.empty(max(2, plan.loss_ws_doubles), dtype=torch.float64)
    args = lambda: (ctypes.byref(plan), ctypes.byref(lay), L._ptr(v), L._ptr(t), v.stride(0), t.stride(0), nat.IN_F32, None, L._ptr(per), L._ptr(tra),
                    L._ptr(loss_ws), 0)
    assert lib.crossclr_step_forward(*args()) == 0
    lay.stash_bytes += 256                                          # tampered
    assert lib.crossclr_step_forward(*args()) == -1 and b"crossclr_step_plan" in lib.crossclr_last_error()
    lay.stash_bytes -= 256
    other = nat.make_plan(48, 24, 1, 0, nat.MODE_FP32)              # another plan's layout
    assert lib.crossclr_step_forward(ctypes.byref(other), *args()[1:]) == -1
    assert lib.crossclr_step_forward(*args()[:9], 0, *args()[10:]) == -1      # transient missing


@pytest.mark.parametrize("B,D,mode,tau,tol", [(150, 32, "bf16", 0.05, 2e-2), (40, 24, "fp32", 0.05, 1e-5), (64, 32, "fp32", 0.004, 1e-5)])
def test_transient_region_is_released_and_a_second_backward_recomputes(B, D, mode, tau, tol, monkeypatch):
    """The autograd function gives the transient region back as soon as nothing reads it: after the forward call with the eager gradient
    product, after the first backward without it -- where a second backward through the same graph recomputes from the persistent region.
    The environment is read by crossclr_step_plan only: changing a knob between forward and backward cannot desynchronise the two calls."""
    v, t = orc.make_inputs("randn", B, D, 7)
    for eager in ("1", "0"):
        monkeypatch.setenv("CROSSCLR_EAGER_BACKWARD", eager)
        monkeypatch.delenv("CROSSCLR_MAX_STASH_GB", raising=False)
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, tau, 0.8, compute_mode=mode)
        ws = loss.grad_fn.ws
        assert L._last_step_saved and (ws.step[2] is None) == (eager == "1")
        monkeypatch.setenv("CROSSCLR_MAX_STASH_GB", "0")        # (would have planned the recomputing layout: the layout in hand still rules)
        loss.backward(retain_graph=True)
        assert ws.step[2] is None and ws.stash is None
        g1, vv.grad = vv.grad.clone(), None
        loss.backward()
        scale = g1.abs().max().item()
        assert (vv.grad - g1).abs().max().item() <= tol * scale
        if eager == "1":
            assert torch.equal(vv.grad, g1)


def test_last_kernel_names_what_the_step_launched():
    """crossclr_last_kernel (ABI 7, reporting aid: bench.py labels its dominant kernel with it): the kernel templates the step's forward and
    gradient product went to, as the launchers recorded them -- here the exact-fp32 plan's generic kernels and a bf16 plan's register-resident ones."""
    lib = nat.library()
    v, t = orc.make_inputs("randn", 40, 24, 3)
    vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="fp32").backward()
    assert lib.crossclr_last_kernel(0).startswith(b"fwd_sums_kernel") and lib.crossclr_last_kernel(1).startswith(b"bwd_")
    v, t = orc.make_inputs("randn", 256, 16, 3)
    vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="bf16").backward()
    assert lib.crossclr_last_kernel(0) == b"fast_fwd_pair_kernel" and lib.crossclr_last_kernel(1).startswith(b"fast_bwd_")
