"""CPU tests of the score-statistics path (max-margin ranking loss, retrieval ranks): the oracle against the golden vectors made
by running the reference's own MaxMargin_coot.forward, and the REAL kernel sources (lane-level emulation) against both."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import crossclr_amd
from crossclr_amd import _native as nat
from oracle import crossclr_oracle as orc
from oracle import ranking_oracle as rk

HERE = os.path.dirname(os.path.abspath(__file__))
IDX = json.load(open(os.path.join(HERE, "golden", "mm_index.json")))


def golden_case(name):
    m = IDX[name]
    im, s = orc.make_inputs(m["kind"], m["B"], m["D"], m["seed"])
    if m["unit_rows"]:
        im, s = torch.nn.functional.normalize(im, dim=1), torch.nn.functional.normalize(s, dim=1)
    return m, im, s, np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("name", sorted(IDX))
def test_oracle_matches_the_reference_golden_vectors(name):
    m, im, s, z = golden_case(name)
    o = rk.max_margin_loss_and_grads(im, s, m["margin"])
    assert np.array_equal(o["loss"].numpy(), z["loss"]) and np.array_equal(o["grad_im"].numpy(), z["grad_im"])
    assert np.array_equal(o["grad_s"].numpy(), z["grad_s"])
    st = rk.max_margin_streaming(im, s, m["margin"])
    scale = max(m["grad_absmax"], 1e-12)
    assert abs(float(st["loss"]) - m["loss"]) <= 1e-5 * max(1.0, abs(m["loss"]))
    assert np.abs(st["grad_im"].numpy() - z["grad_im"]).max() <= 1e-5 * scale + 1e-9


@pytest.mark.skipif(not os.path.exists("/root/reference/trainer/loss.py"), reason="needs the reference (build container only)")
def test_golden_recipe_regenerates_bit_for_bit():
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_golden_ranking.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.fixture(scope="module")
def emulated_library():
    from emu import build_emu
    nat.use_library_for_testing(build_emu.build())
    yield
    nat.use_library_for_testing(None)


@pytest.mark.parametrize("name", ["mm_b8_d16_s1", "mm_b70_d48_s2", "mm_raw_b32_d24_s3", "mm_m0_b64_d32_s6"])
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_emulated_kernels_match_the_reference_golden_vectors(emulated_library, name, mode):
    m, im, s, z = golden_case(name)
    a, b = im.clone().requires_grad_(True), s.clone().requires_grad_(True)
    crit = crossclr_amd.MaxMargin_coot(use_cuda=False, margin=m["margin"], compute_mode=mode)
    loss = crit(a, b)
    (3.0 * loss).backward()
    assert loss.dim() == 0 and loss.dtype == im.dtype
    if mode == "fp32":
        assert abs(loss.item() - m["loss"]) <= 2e-6 * max(1.0, abs(m["loss"]))
        # a hinge exactly at its kink may be counted on either side by differently-ordered fp32 sums: each flips one row of weight
        # 1/B^2; allow a handful of such rows
        err = np.abs(a.grad.numpy() / 3.0 - z["grad_im"]).max(axis=1)
        assert (err > 2e-6 * max(m["grad_absmax"], 1e-9)).sum() <= 2
    else:   # bf16 operands move scores by ~4e-3: hinges near the kink flip -> compare with the closed form on the ROUNDED operands
        ar, br = im.bfloat16().float(), s.bfloat16().float()
        st = rk.max_margin_streaming(ar, br, m["margin"])
        assert abs(loss.item() - float(st["loss"])) <= 2e-5 * max(1.0, abs(float(st["loss"])))
        # gradient = (W x_rounded - c * partner_fp32) / B^2: rebuild with the fp32 partner term like the kernels do
        B = im.shape[0]
        S = ar.double() @ br.double().t()
        d = S.diag()
        off = ~torch.eye(B, dtype=torch.bool)
        Wm = (((m["margin"] + S - d[:, None]) > 0) & off).double() + (((m["margin"] + S - d[None, :]) > 0) & off).double()
        c = st["active_im"].double() + st["active_s"].double()
        g_im = (Wm @ br.double() - c[:, None] * s.double()) / (B * B)
        err = (a.grad.double() / 3.0 - g_im).abs().max(dim=1).values
        assert (err > 1e-4 * max(float(g_im.abs().max()), 1e-9)).sum() <= 2


@pytest.mark.parametrize("B,D,normalize", [(70, 48, True), (130, 24, False), (8, 16, True)])
def test_emulated_retrieval_ranks(emulated_library, B, D, normalize):
    v, t = orc.make_inputs("cluster", B, D, 9)
    t = t + 0.05 * torch.randn(B, D, generator=torch.Generator().manual_seed(1))
    got = crossclr_amd.retrieval_ranks(v, t, normalize=normalize, compute_mode="fp32")
    ref = rk.retrieval_ranks_dense(v, t, normalize=normalize)
    # fp32 scores vs the float64 oracle: a candidate within rounding of the partner's score may fall on either side
    S = ref["scores"]
    d = S.diag()
    for key, ties in (("v2t_ranks", ((S - d[:, None]).abs() < 1e-5).sum(1) - 1), ("t2v_ranks", ((S - d[None, :]).abs() < 1e-5).sum(0) - 1)):
        assert ((got[key] - ref[key]).abs() <= ties).all(), key
    assert got["v2t"].shape == (5,) and 0.0 <= float(got["v2t"][0]) <= float(got["v2t"][2]) <= 1.0
    assert torch.allclose(got["t2v"][:3], ref["t2v"][:3], atol=3.0 / B)


@pytest.mark.parametrize("B,D,mode", [(70, 48, "fp32"), (150, 130, "fp32"), (130, 72, "bf16")])
def test_backward_from_the_saved_hinge_mask_is_the_recomputing_backward(emulated_library, B, D, mode, monkeypatch):
    """With a backward to follow, crossclr_score_rows_save also leaves every pair's number of active hinges (one byte per pair) and the
    backward is one product with that mask (crossclr_maxmargin_backward_saved) instead of a second evaluation of the scores: same loss bits,
    same gradient bits as the recomputing pair (the scores come from the same MFMA sequence), ragged batches and padding included."""
    g = torch.Generator().manual_seed(B)
    im, s = torch.randn(B, D, generator=g), torch.randn(B, D, generator=g)
    if mode == "bf16":
        im, s = torch.nn.functional.normalize(im, dim=1), torch.nn.functional.normalize(s, dim=1)

    def run():
        a, b = im.clone().requires_grad_(True), s.clone().requires_grad_(True)
        loss = crossclr_amd.max_margin_loss(a, b, 0.1, compute_mode=mode)
        saved = loss.grad_fn.sc.mask is not None
        (1.5 * loss).backward()
        return loss.item(), a.grad, b.grad, saved
    l1, ga1, gb1, saved1 = run()
    monkeypatch.setenv("CROSSCLR_MAXMARGIN_SAVE", "0")
    l0, ga0, gb0, saved0 = run()
    assert saved1 and not saved0
    assert l1 == l0 and torch.equal(ga1, ga0) and torch.equal(gb1, gb0)
    monkeypatch.delenv("CROSSCLR_MAXMARGIN_SAVE")
    with torch.no_grad():      # nothing to save without a backward
        assert crossclr_amd.max_margin_loss(im, s, 0.1, compute_mode=mode).item() == l1
    monkeypatch.setenv("CROSSCLR_DISABLE_SYMMETRIC", "1")      # the two-pass evaluation writes no mask: the library declines, the module recomputes
    l2, ga2, gb2, saved2 = run()
    assert not saved2 and torch.allclose(ga2, ga1, rtol=1e-5, atol=1e-7)
