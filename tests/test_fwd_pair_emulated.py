"""CPU tests (host emulation of the real kernel sources): the symmetric forward with one unbroken MFMA stream per wave
(fast_fwd_pair_kernel, csrc/crossclr_kernels_symp.h) against the kernel it replaces for whole batches (fast_fwd_pipe_kernel):
same work list, same summation order -> the loss, the saved exponentials and therefore the gradients must agree BIT FOR BIT.
The emulation runs the same index math, ring protocol and barrier placement as the GPU build; waits and hazards are the GPU tests' part
(tests/test_gpu_fwd_pair.py)."""
import pytest
import torch

import crossclr_amd
from crossclr_amd import _native as nat
from oracle import crossclr_oracle as orc


@pytest.fixture(scope="module", autouse=True)
def emulated_library():
    from emu import build_emu
    nat.use_library_for_testing(build_emu.build())
    assert nat.backend() == "emu-host"
    yield
    nat.use_library_for_testing(None)


def step(v, t, grad=True):
    vv, tt = v.clone().requires_grad_(grad), t.clone().requires_grad_(grad)
    if not grad:
        with torch.no_grad():
            return crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="bf16"), None, None
    loss = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="bf16")
    loss.backward()
    return loss, vv.grad, tt.grad


# (B, D, CROSSCLR_FWD_BLOCKS): one thread block walks every row block (long runs of pipelined pairs, several segments per block);
# a few blocks (ranges that start and end inside a row block's diagonal tiles); the default (ranges of 1-3 tiles: every seam exercised)
CASES = [(128, 16, 1), (256, 16, 1), (256, 200, 3), (384, 40, 2), (384, 16, 5), (256, 24, 0), (128, 300, 0), (256, 400, 2),
         # wide operands (512 < D <= 1024): 128-row blocks, one 32-row half per wave, the column tile in two ring stages
         (128, 600, 1), (256, 1000, 2), (256, 700, 0)]


@pytest.mark.parametrize("B,D,blocks", CASES)
def test_pair_forward_is_bit_identical_to_the_pipe_forward(B, D, blocks, monkeypatch):
    if blocks:
        monkeypatch.setenv("CROSSCLR_FWD_BLOCKS", str(blocks))
    plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16)
    assert plan.fast_path == 1 and plan.Dpad <= 1024 and plan.bpad == B
    v, t = orc.make_inputs("randn", B, D, 41)
    monkeypatch.delenv("CROSSCLR_FWD_PAIR", raising=False)
    ln, gvn, gtn = step(v, t)
    lf, _, _ = step(v, t, grad=False)                 # (without save: the ST = false instantiation)
    monkeypatch.setenv("CROSSCLR_FWD_PAIR", "0")
    lo, gvo, gto = step(v, t)
    assert ln.item() == lo.item() and lf.item() == lo.item()
    assert torch.equal(gvn, gvo) and torch.equal(gtn, gto)
    model = float(orc.bf16_operand_model_loss(v, t, 0.05, 0.8))
    assert abs(ln.item() - model) <= 2e-6 * max(1.0, abs(model))


@pytest.mark.parametrize("B,D,blocks", [(128, 600, 1), (256, 1000, 2), (256, 700, 0)])
def test_pair_forward_with_sample_weights_is_bit_identical(B, D, blocks, monkeypatch):
    """Per-sample weights (negative scales k, loss weights omega) on the wide instantiations <DK, ST, 1, 2, KIND, SW = true>: the tile's column
    scales travel in registers from the tile's step to its epilogue; inter-modal tiles multiply by 1.0."""
    if blocks:
        monkeypatch.setenv("CROSSCLR_FWD_BLOCKS", str(blocks))
    v, t = orc.make_inputs("randn", B, D, 43)
    g = torch.Generator().manual_seed(B + D)
    keep = lambda: (torch.rand(B, generator=g) > 0.3).float() * (0.5 + torch.rand(B, generator=g))
    kw = dict(negative_scale=(keep(), keep()), loss_weight=(torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5))

    def wstep():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="bf16", **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    monkeypatch.delenv("CROSSCLR_FWD_PAIR", raising=False)
    ln, gvn, gtn = wstep()
    monkeypatch.setenv("CROSSCLR_FWD_PAIR", "0")
    lo, gvo, gto = wstep()
    assert ln == lo and torch.equal(gvn, gvo) and torch.equal(gtn, gto)
