"""CPU tests (host emulation): the rectangular (KIND 2) and pair (KIND 3) launches of sharded runs on fast_fwd_pair_kernel against
fast_fwd_pipe_kernel, through the C-ABI entry points a rank calls for the blocks against OTHER ranks (crossclr_forward_pairs,
crossclr_forward_rect_save, crossclr_forward_w with a skipped rank): partial row sums, column sums for the partner and the rectangular
stash must agree BIT FOR BIT.  (tests/test_gpu_fwd_pair.py repeats it on the MI355X.)"""
import ctypes

import pytest
import torch

from crossclr_amd import _native as nat
from crossclr_amd import loss as L
from oracle import crossclr_oracle as orc


@pytest.fixture(scope="module", autouse=True)
def emulated_library():
    from emu import build_emu
    nat.use_library_for_testing(build_emu.build())
    yield
    nat.use_library_for_testing(None)


def rect_outputs(world, rank, b, D, dev="cpu", stream=0, weighted=False):
    """Every rectangular launch shape the sharded host path uses, for `rank` of `world`: returns the tensors they wrote."""
    lib = nat.library()
    plan = nat.make_plan(b, D, world, rank, nat.MODE_BF16)
    pp, p = ctypes.byref(plan), L._ptr
    f32 = dict(dtype=torch.float32, device=dev)
    xs = []
    for r in range(world):
        v, t = orc.make_inputs("randn", b, D, 100 + r)
        v, t = v.to(dev), t.to(dev)
        xh = torch.empty(plan.operand_bytes, dtype=torch.uint8, device=dev)
        inv_norm, diag = torch.empty(2 * plan.bpad, **f32), torch.empty(plan.bpad, **f32)      # (named: a bare address does not keep a tensor alive)
        nat.check(lib.crossclr_normalize(pp, p(v), p(t), v.stride(0), t.stride(0), nat.IN_F32, p(xh), p(inv_norm), p(diag), stream))
        xs.append(xh)
    xcols = torch.cat(xs)
    xhat = xs[rank]
    n2 = 2 * plan.bpad
    out = {}
    sw = None
    if weighted:      # negative scales of every rank's rows (statistics layout), some of them zero (pruned)
        gk = torch.Generator().manual_seed(9)
        k_all = ((torch.rand(world * n2, generator=gk) > 0.3).float() * (0.5 + torch.rand(world * n2, generator=gk))).to(dev)
        k_rows = k_all[rank * n2:(rank + 1) * n2]
        sw = L._sw(k_rows, k_all, None)
        out["_keepalive"] = k_all
    npairs = (world - 1) // 2
    part = torch.zeros(plan.fwd_ws_floats, **f32)
    if npairs:
        first = (rank + 1) % world
        colsum = torch.zeros(npairs * n2, **f32)
        nat.check(lib.crossclr_forward_pairs(pp, p(xhat), p(xcols), first, npairs, 0.05, 0.8, sw, p(part), plan.fwd_slots, p(colsum), stream))
        out["pairs_part"], out["pairs_colsum"] = part.clone(), colsum.clone()
        nb = lib.crossclr_rect_stash_bytes(pp, npairs)
        if nb:
            st = torch.zeros(nb, dtype=torch.uint8, device=dev)
            part.zero_(); colsum.zero_()
            nat.check(lib.crossclr_forward_rect_save(pp, p(xhat), p(xcols), first, npairs, 1, 0.05, 0.8, sw, p(part), plan.fwd_slots, p(colsum),
                                                     p(st), stream))
            out["rect_part"], out["rect_colsum"], out["rect_stash"] = part.clone(), colsum.clone(), st
    if world % 2 == 0:
        opp = (rank + world // 2) % world
        nb = lib.crossclr_rect_stash_bytes(pp, 1)
        st = torch.zeros(max(nb, 1), dtype=torch.uint8, device=dev)
        part.zero_()
        nat.check(lib.crossclr_forward_rect_save(pp, p(xhat), p(xcols), opp, 1, 0, 0.05, 0.8, sw, p(part), 2 * plan.fwd_slots, None, p(st), stream))
        out["opp_part"], out["opp_stash"] = part.clone(), st
    part.zero_()       # every other rank, the own one skipped (the plain sharded scheme)
    nat.check(lib.crossclr_forward_w(pp, p(xhat), p(xcols), world, 0, rank, 0.05, 0.8, sw, p(part), plan.fwd_slots, stream))
    out["skip_part"] = part.clone()
    return out


@pytest.mark.parametrize("world,rank,b,D", [(3, 1, 128, 16), (4, 3, 128, 40), (2, 0, 256, 16), (5, 2, 128, 600)])
def test_rectangular_and_pair_launches_are_bit_identical(world, rank, b, D, monkeypatch):
    monkeypatch.delenv("CROSSCLR_FWD_PAIR", raising=False)
    new = rect_outputs(world, rank, b, D)
    monkeypatch.setenv("CROSSCLR_FWD_PAIR", "0")
    old = rect_outputs(world, rank, b, D)
    assert set(new) == set(old) and len(new) >= 2
    for k in new:
        assert torch.equal(new[k], old[k]), k
    assert any(float(new[k].float().abs().sum()) > 0 for k in new if k.endswith("part"))


@pytest.mark.parametrize("world,rank,b,D", [(3, 1, 128, 600), (4, 2, 128, 1000)])
def test_weighted_rectangular_and_pair_launches_are_bit_identical(world, rank, b, D, monkeypatch):
    monkeypatch.delenv("CROSSCLR_FWD_PAIR", raising=False)
    new = rect_outputs(world, rank, b, D, weighted=True)
    monkeypatch.setenv("CROSSCLR_FWD_PAIR", "0")
    old = rect_outputs(world, rank, b, D, weighted=True)
    for k in new:
        assert torch.equal(new[k], old[k]), k
