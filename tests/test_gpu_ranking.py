"""MI355X tests of the score-statistics path through the C-ABI: the max-margin ranking loss against the golden vectors made by
the reference's own MaxMargin_coot.forward (trainer/loss.py:29-41) and, at sizes the reference cannot hold, against the
float64 closed form; retrieval ranks against the dense float64 definition."""
import json
import os

import numpy as np
import pytest
import torch

import crossclr_amd
from crossclr_amd import _native as nat
from oracle import crossclr_oracle as orc
from oracle import ranking_oracle as rk

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
IDX = json.load(open(os.path.join(HERE, "golden", "mm_index.json")))


def golden_case(name):
    m = IDX[name]
    im, s = orc.make_inputs(m["kind"], m["B"], m["D"], m["seed"])
    if m["unit_rows"]:
        im, s = torch.nn.functional.normalize(im, dim=1), torch.nn.functional.normalize(s, dim=1)
    return m, im, s, np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("name", sorted(IDX))
def test_max_margin_matches_the_reference_golden_vectors(name):
    assert nat.backend() == "hip-gfx950"
    m, im, s, z = golden_case(name)
    a, b = im.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    loss = crossclr_amd.MaxMargin_coot(use_cuda=True, margin=m["margin"], compute_mode="fp32")(a, b)
    (2.0 * loss).backward()
    assert loss.is_cuda and loss.dim() == 0 and loss.dtype == torch.float32
    assert abs(loss.item() - m["loss"]) <= 2e-6 * max(1.0, abs(m["loss"]))
    scale = max(m["grad_absmax"], 1e-9)
    for got, want in ((a.grad, z["grad_im"]), (b.grad, z["grad_s"])):
        err = np.abs(got.cpu().numpy() / 2.0 - want).max(axis=1)
        assert (err > 2e-6 * scale).sum() <= 2      # a hinge exactly at its kink may fall on either side (one row each)


@pytest.mark.parametrize("mode,B,D", [("fp32", 2048, 512), ("bf16", 2048, 512), ("bf16", 1000, 300), ("fp32", 333, 1100)])
def test_max_margin_large_batches_match_the_float64_closed_form(mode, B, D):
    im, s = orc.make_inputs("cluster", B, D, 13)
    im, s = torch.nn.functional.normalize(im, dim=1), torch.nn.functional.normalize(s, dim=1)
    a, b = im.cuda().requires_grad_(True), s.cuda().requires_grad_(True)
    loss = crossclr_amd.max_margin_loss(a, b, 0.1, compute_mode=mode)
    loss.backward()
    src = (im.bfloat16().float(), s.bfloat16().float()) if mode == "bf16" else (im, s)   # what the MFMAs were fed
    st = rk.max_margin_streaming(src[0], src[1], 0.1)
    assert abs(loss.item() - float(st["loss"])) <= 1e-4 * max(1.0, abs(float(st["loss"])))
    # gradients: flipped hinges at the kink (|margin + S - d| within rounding) move single weights by 1/B^2 -> bound the damage
    ref_im = rk.max_margin_streaming(im, s, 0.1)["grad_im"]
    scale = float(ref_im.abs().max())
    err = (a.grad.double().cpu() - ref_im).abs().max().item()
    assert err <= (5e-2 if mode == "bf16" else 2e-3) * scale
    assert torch.isfinite(b.grad).all()


@pytest.mark.parametrize("B,D,normalize,mode", [(4096, 512, True, "fp32"), (1000, 300, False, "fp32"), (4096, 512, True, "bf16")])
def test_retrieval_ranks_match_the_dense_definition(B, D, normalize, mode):
    v, t = orc.make_inputs("cluster", B, D, 21)
    t = t + 0.3 * torch.randn(B, D, generator=torch.Generator().manual_seed(2))
    got = crossclr_amd.retrieval_ranks(v.cuda(), t.cuda(), normalize=normalize, compute_mode=mode)
    ref = rk.retrieval_ranks_dense(v, t, normalize=normalize)
    S = ref["scores"]
    d = S.diag()
    tol = 1e-5 * float(S.abs().max()) if mode == "fp32" else 2e-2 * float(S.abs().max())
    for key, near in (("v2t_ranks", ((S - d[:, None]).abs() < tol).sum(1) - 1), ("t2v_ranks", ((S - d[None, :]).abs() < tol).sum(0) - 1)):
        assert ((got[key].cpu() - ref[key]).abs() <= near).all(), key     # only candidates within rounding of the partner may swap
    if mode == "fp32":
        assert torch.allclose(got["v2t"].cpu()[:3], ref["v2t"][:3], atol=2e-3) and torch.allclose(got["t2v"].cpu()[:3], ref["t2v"][:3], atol=2e-3)
    assert got["v2t_ranks"].is_cuda and got["v2t_ranks"].dtype == torch.int64


def test_retrieval_of_identical_sets_is_perfect_and_timed():
    B, D = 8192, 512
    v, _ = orc.make_inputs("randn", B, D, 3)
    vd = v.cuda()
    got = crossclr_amd.retrieval_ranks(vd, vd.clone(), compute_mode="fp32")
    assert int(got["v2t_ranks"].max()) == 0 and int(got["t2v_ranks"].max()) == 0 and float(got["v2t"][0]) == 1.0
    for mode in ("fp32", "bf16"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        crossclr_amd.retrieval_ranks(vd, vd, compute_mode=mode)
        e0.record()
        for _ in range(5):
            crossclr_amd.retrieval_ranks(vd, vd, compute_mode=mode)
        e1.record()
        torch.cuda.synchronize()
        print(f"retrieval ranks B={B} D={D} {mode}: {e0.elapsed_time(e1) / 5:.3f} ms")


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float64])
def test_max_margin_input_dtypes_and_strided_rows(dtype):
    """Inputs in any float dtype of the reference, rows taken as a strided view: loss in the input dtype (like the reference's
    eager ops), gradients in the input dtype, values within the dtype's resolution of the float64 closed form on the same rows."""
    g = torch.Generator().manual_seed(8)
    big_a, big_b = torch.randn(200, 2 * 80, generator=g), torch.randn(200, 2 * 80, generator=g)
    a = torch.nn.functional.normalize(big_a[:, :80], dim=1).to(dtype).cuda()
    b = torch.nn.functional.normalize(big_b[:, :80], dim=1).to(dtype).cuda()
    wide_a = torch.zeros(200, 160, dtype=dtype, device="cuda"); wide_a[:, :80] = a
    av = wide_a[:, :80].requires_grad_(True)          # row stride 160
    bv = b.clone().requires_grad_(True)
    loss = crossclr_amd.max_margin_loss(av, bv, 0.1, compute_mode="fp32")
    loss.backward()
    assert loss.dtype == dtype and av.grad.dtype == dtype and av.grad.shape == av.shape
    st = rk.max_margin_streaming(a.cpu().double(), b.cpu().double(), 0.1)
    eps = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2, torch.float64: 1e-6}[dtype]
    assert abs(loss.item() - float(st["loss"])) <= eps * max(1.0, abs(float(st["loss"])))
    scale = float(st["grad_im"].abs().max())
    assert (av.grad.double().cpu() - st["grad_im"]).abs().max().item() <= max(eps, 2e-3) * scale * 4


@pytest.mark.parametrize("mode,B,D", [("fp32", 8192, 512), ("bf16", 8192, 512), ("fp32", 1000, 300), ("fp32", 333, 1100), ("bf16", 2050, 96)])
def test_backward_from_the_saved_hinge_mask_is_the_recomputing_backward_on_the_device(mode, B, D, monkeypatch):
    """crossclr_score_rows_save + crossclr_maxmargin_backward_saved (one byte per pair: its number of active hinges; the backward is one
    product with the mask) against the recomputing pair: same loss bits, same gradient bits -- the headline batch, ragged batches, wide rows."""
    im, s = orc.make_inputs("cluster", B, D, 17)
    im, s = torch.nn.functional.normalize(im, dim=1).cuda(), torch.nn.functional.normalize(s, dim=1).cuda()

    def run():
        a, b = im.clone().requires_grad_(True), s.clone().requires_grad_(True)
        loss = crossclr_amd.max_margin_loss(a, b, 0.1, compute_mode=mode)
        saved = loss.grad_fn.sc.mask is not None
        loss.backward()
        return loss.item(), a.grad, b.grad, saved
    l1, ga1, gb1, saved1 = run()
    monkeypatch.setenv("CROSSCLR_MAXMARGIN_SAVE", "0")
    l0, ga0, gb0, saved0 = run()
    assert saved1 and not saved0 and l1 == l0
    assert torch.equal(ga1, ga0) and torch.equal(gb1, gb0)
