"""GPU tests: fast_fwd_pair_kernel (csrc/crossclr_kernels_symp.h: the symmetric forward with one unbroken MFMA stream per wave) against
fast_fwd_pipe_kernel, the kernel it replaces for whole batches -- same work list, same summation order: loss, saved exponentials and
gradients BIT FOR BIT, launch after launch (hand-counted lgkmcnt / vmcnt waits, the mid-tile barrier, asm MFMAs with VGPR accumulators:
everything the host emulation cannot see).  Reference goldens: tests/test_gpu_parity.py runs them through this kernel by default."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _step(crit, v, t, grad=True):
    if not grad:
        with torch.no_grad():
            return crit(v, t).item(), None, None
    vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    loss = crit(vv, tt)
    loss.backward()
    return loss.item(), vv.grad.clone(), tt.grad.clone()


SHAPES = [(128, 128), (256, 100), (384, 256), (512, 384), (1024, 512), (1152, 200), (2048, 512), (4096, 384), (8192, 512), (3200, 512),
          (128, 1024), (384, 600), (1024, 768), (2048, 1024), (8192, 1024), (4224, 900)]      # wide operands: the tile in two ring stages


@pytest.mark.parametrize("B,D", SHAPES)
def test_pair_forward_is_bit_identical_to_the_pipe_forward(B, D, monkeypatch):
    import crossclr_amd
    from crossclr_amd import _native as nat
    assert nat.backend() == "hip-gfx950"
    g = torch.Generator().manual_seed(B + D)
    v = torch.randn(B, D, generator=g).cuda()
    t = (0.6 * v.cpu() + torch.randn(B, D, generator=g)).cuda()       # correlated pairs: exponentials of mixed magnitude
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.05, 0.8, compute_mode="bf16").cuda()
    monkeypatch.setenv("CROSSCLR_FWD_PAIR", "0")
    lo, gvo, gto = _step(crit, v, t)
    lfo, _, _ = _step(crit, v, t, grad=False)
    monkeypatch.delenv("CROSSCLR_FWD_PAIR")
    for rep in range(4):
        ln, gvn, gtn = _step(crit, v, t)
        assert ln == lo, (rep, ln, lo)
        assert torch.equal(gvn, gvo) and torch.equal(gtn, gto), rep
        lfn, _, _ = _step(crit, v, t, grad=False)
        assert lfn == lfo
    torch.cuda.synchronize()


def test_pair_forward_many_launches_headline_shape():
    """BASELINE config 3's shape, 200 steps: every loss and every gradient identical to the first."""
    import crossclr_amd
    g = torch.Generator().manual_seed(1234)
    v, t = torch.randn(8192, 512, generator=g).cuda(), torch.randn(8192, 512, generator=g).cuda()
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
    l0, gv0, gt0 = _step(crit, v, t)
    bad = 0
    for _ in range(200):
        l, gv, gt = _step(crit, v, t)
        bad += int(l != l0) + int(not torch.equal(gv, gv0)) + int(not torch.equal(gt, gt0))
    assert bad == 0
    assert abs(l0 - 10.627098744839678) <= 1e-3        # tests/golden/index.json: g7_b8192_d512_s1234


@pytest.mark.parametrize("world,rank,b,D", [(3, 1, 128, 128), (4, 3, 256, 200), (2, 0, 1024, 512), (8, 4, 1024, 512), (5, 2, 384, 1024), (8, 7, 2048, 256)])
def test_rectangular_and_pair_launches_are_bit_identical(world, rank, b, D, monkeypatch):
    """The blocks a rank of a sharded run evaluates against OTHER ranks (crossclr_forward_pairs, crossclr_forward_rect_save, crossclr_forward_w
    with the own rank skipped) on fast_fwd_pair_kernel<..., KIND 2 / 3> against fast_fwd_pipe_kernel: partial row sums, the partner's column
    sums and the rectangular stash bit for bit, twice."""
    from test_fwd_pair_rect_emulated import rect_outputs
    stream = torch.cuda.current_stream().cuda_stream
    monkeypatch.setenv("CROSSCLR_FWD_PAIR", "0")
    old = rect_outputs(world, rank, b, D, dev="cuda", stream=stream)
    monkeypatch.delenv("CROSSCLR_FWD_PAIR")
    for _ in range(2):
        new = rect_outputs(world, rank, b, D, dev="cuda", stream=stream)
        assert set(new) == set(old) and len(new) >= 2
        for k in new:
            assert torch.equal(new[k], old[k]), k
    torch.cuda.synchronize()


@pytest.mark.parametrize("B,D", [(1024, 1024), (2048, 768), (8192, 1024)])
def test_weighted_pair_forward_is_bit_identical(B, D, monkeypatch):
    """Sample weights on the wide instantiations (BASELINE config 5's width): column scales by vmcnt-counted asm loads, blended to 1.0 for
    inter-modal tiles -- loss and gradients bit for bit against fast_fwd_pipe_kernel<..., SW>."""
    import crossclr_amd
    g = torch.Generator().manual_seed(B + D)
    v = torch.randn(B, D, generator=g).cuda()
    t = (0.6 * v.cpu() + torch.randn(B, D, generator=g)).cuda()
    keep = lambda: ((torch.rand(B, generator=g) > 0.3).float() * (0.5 + torch.rand(B, generator=g))).cuda()
    kw = dict(negative_scale=(keep(), keep()), loss_weight=((torch.rand(B, generator=g) + 0.5).cuda(), (torch.rand(B, generator=g) + 0.5).cuda()))

    def wstep():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="bf16", **kw)
        loss.backward()
        return loss.item(), vv.grad.clone(), tt.grad.clone()
    monkeypatch.setenv("CROSSCLR_FWD_PAIR", "0")
    lo, gvo, gto = wstep()
    monkeypatch.delenv("CROSSCLR_FWD_PAIR")
    for _ in range(3):
        ln, gvn, gtn = wstep()
        assert ln == lo and torch.equal(gvn, gvo) and torch.equal(gtn, gto)


@pytest.mark.parametrize("world,rank,b,D", [(3, 1, 256, 1024), (8, 4, 1024, 1024), (4, 2, 384, 768)])
def test_weighted_rectangular_and_pair_launches_are_bit_identical(world, rank, b, D, monkeypatch):
    from test_fwd_pair_rect_emulated import rect_outputs
    stream = torch.cuda.current_stream().cuda_stream
    monkeypatch.setenv("CROSSCLR_FWD_PAIR", "0")
    old = rect_outputs(world, rank, b, D, dev="cuda", stream=stream, weighted=True)
    monkeypatch.delenv("CROSSCLR_FWD_PAIR")
    for _ in range(2):
        new = rect_outputs(world, rank, b, D, dev="cuda", stream=stream, weighted=True)
        for k in new:
            assert torch.equal(new[k], old[k]), k
    torch.cuda.synchronize()
