"""GPU tests of the two step calls as the module drives them (C-ABI 7): the workspace is a persistent and a transient region, the transient one
(saved exponentials, operand copies: most of the bytes) goes back to the allocator as soon as nothing reads it, a second backward through the
same graph still works, and the library reports which kernels it launched."""
import ctypes

import pytest
import torch

import crossclr_amd
from crossclr_amd import _native as nat
from crossclr_amd import loss as L
from oracle import crossclr_oracle as orc

pytestmark = pytest.mark.gpu


def _inputs(B, D, seed=3):
    v, t = orc.make_inputs("randn", B, D, seed)
    return v.cuda(), t.cuda()


@pytest.mark.parametrize("eager", ["1", "0"])
def test_transient_region_goes_back_to_the_allocator(eager, monkeypatch):
    """B = 8192, D = 512, bf16: 0.30 GB of the step's 0.37 GB are transient.  With the eager gradient product (default) they are released when the
    forward call returns -- the loss tensor pins 67 MB, not 370 --; without it when the first backward has been enqueued."""
    monkeypatch.setenv("CROSSCLR_EAGER_BACKWARD", eager)
    v, t = _inputs(8192, 512)
    vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
    crit(vv, tt).backward()          # (warm: plans, self-test of the fragment-major kernels, allocator)
    vv.grad = tt.grad = None
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    loss = crit(vv, tt)
    lay = loss.grad_fn.ws.step[0]
    held = torch.cuda.memory_allocated() - base
    assert lay.saved == 1 and lay.transient_bytes > 250 * 2 ** 20 and lay.persistent_bytes < 80 * 2 ** 20
    if eager == "1":
        assert loss.grad_fn.ws.step[2] is None and held < lay.persistent_bytes + 2 ** 20, held
    else:
        assert loss.grad_fn.ws.step[2] is not None and held >= lay.total_bytes
    loss.backward(retain_graph=True)
    assert loss.grad_fn.ws.step[2] is None
    held = torch.cuda.memory_allocated() - base - vv.grad.numel() * 4 * 2
    assert held < lay.persistent_bytes + 2 ** 20, held
    g1, vv.grad = vv.grad.clone(), None
    loss.backward()                  # second backward: the finish kernel alone (eager) / the recomputing product from the persistent region
    scale = g1.abs().max().item()
    assert (vv.grad - g1).abs().max().item() <= (0.0 if eager == "1" else 2e-2) * scale


def test_last_kernel_and_layout_at_the_headline_shape():
    v, t = _inputs(8192, 512)
    vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()(vv, tt).backward()
    torch.cuda.synchronize()
    lib = nat.library()
    assert lib.crossclr_last_kernel(0) == b"fast_fwd_pair_kernel" and lib.crossclr_last_kernel(1) == b"fast_bwd_xfp_kernel"
    assert L._last_step_backward_kernel == 3
    vv.grad = tt.grad = None
    crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="fp32").cuda()(vv, tt).backward()
    torch.cuda.synchronize()
    assert lib.crossclr_last_kernel(0).startswith(b"fwd_sums_kernel") and lib.crossclr_last_kernel(1).startswith(b"bwd_saved32_kernel")


def test_step_calls_through_the_c_abi_with_one_allocation():
    """The two regions may be one allocation (transient = persistent + persistent_bytes), and a tampered layout is refused before anything is launched."""
    lib = nat.library()
    v, t = _inputs(1000, 300, 5)
    plan = nat.make_plan(1000, 300, 1, 0, nat.MODE_BF16)
    lay = nat.StepLayout()
    nat.check(lib.crossclr_step_plan(ctypes.byref(plan), 0.03, 0.8, 0, 0, ctypes.byref(lay)))
    slab = torch.empty(lay.total_bytes, dtype=torch.uint8, device="cuda")
    loss_ws = torch.empty(max(2, plan.loss_ws_doubles), dtype=torch.float64, device="cuda")
    stream = L._stream_for(v)
    per, tra = slab.data_ptr(), slab.data_ptr() + lay.persistent_bytes
    nat.check(lib.crossclr_step_forward(ctypes.byref(plan), ctypes.byref(lay), L._ptr(v), L._ptr(t), v.stride(0), t.stride(0), nat.IN_F32, None, per, tra,
                                        L._ptr(loss_ws), stream))
    scratch = torch.empty(lay.backward_scratch_bytes, dtype=torch.uint8, device="cuda")
    go = torch.ones(1, dtype=torch.float64, device="cuda")
    gv, gt = torch.empty_like(v), torch.empty_like(t)
    nat.check(lib.crossclr_step_backward(ctypes.byref(plan), ctypes.byref(lay), L._ptr(v), L._ptr(t), v.stride(0), t.stride(0), nat.IN_F32, None, per, tra,
                                         L._ptr(scratch), L._ptr(go), L._ptr(gv), L._ptr(gt), gv.stride(0), gt.stride(0), stream))
    vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    want = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()(vv, tt)
    want.backward()
    assert loss_ws[1].item() == want.item() and torch.equal(gv, vv.grad) and torch.equal(gt, tt.grad)
    lay.xhat += 256
    assert lib.crossclr_step_backward(ctypes.byref(plan), ctypes.byref(lay), L._ptr(v), L._ptr(t), v.stride(0), t.stride(0), nat.IN_F32, None, per, tra,
                                      L._ptr(scratch), L._ptr(go), L._ptr(gv), L._ptr(gt), gv.stride(0), gt.stride(0), stream) == -1
