"""CPU tests: the REAL kernel sources (crossmodal-contrastive-learning_amd/csrc) executed lane by
lane on host threads through the emulation shim in tests/emu/ and checked against the reference's
golden vectors.  Shapes are tiny (the emulation runs one OS thread per GPU lane); the same code at
full size is covered by the `-m gpu` tests.  The emulated library is injected explicitly here --
the product binding never selects it."""
import ctypes

import numpy as np
import pytest
import torch

import crossclr_amd
from conftest import golden_arrays, golden_index, golden_inputs
from crossclr_amd import _native as nat
from oracle import crossclr_oracle as orc

IDX = golden_index()


@pytest.fixture(scope="module", autouse=True)
def emulated_library():
    from emu import build_emu
    nat.use_library_for_testing(build_emu.build())
    assert nat.backend() == "emu-host"
    yield
    nat.use_library_for_testing(None)


def run(v, t, tau, w, mode, scale=1.0):
    vv = v.clone().requires_grad_(True)
    tt = t.clone().requires_grad_(True)
    loss = crossclr_amd.crossclr_loss(vv, tt, tau, w, compute_mode=mode)
    (loss * scale).backward()
    return loss, vv.grad, tt.grad


TINY = ["g2_b8_d16_s1", "g4_b16_d32_s3_float32", "g4_b16_d32_s3_float64", "g4_b16_d32_s3_float16",
        "g4_b16_d32_s3_bfloat16", "g5_zero_row_b16_d32", "g5_b1_d32", "g5_w0_tau01_b16_d32", "g5_tau01_b16_d32",
        "g5_tau002_b32_d64", "g5_tau0005_b32_d64", "g5_tau0002_b64_d128", "g5_tau0005_w4_b48_d40", "g5_tiny_row_b16_d32"]


@pytest.mark.parametrize("name", TINY)
def test_fp32_kernels_match_reference(name):
    m = IDX[name]
    v, t = golden_inputs(m)
    loss, gv, gt = run(v, t, m["temperature"], m["negative_weight"], "fp32")
    assert loss.dtype == torch.float64 and loss.dim() == 0
    assert gv.dtype == v.dtype
    half_in = m["dtype"] in ("float16", "bfloat16")
    assert abs(loss.item() - m["loss"]) <= (5e-3 if half_in else 2e-5 * max(1.0, abs(m["loss"])))
    arr = golden_arrays(name)
    scale = max(m["grad_v_absmax"], m["grad_t_absmax"])
    tol = (2e-2 if half_in else 2e-4) * scale
    assert np.abs(gv.double().numpy() - arr["grad_v"].astype(np.float64)).max() <= tol
    assert np.abs(gt.double().numpy() - arr["grad_t"].astype(np.float64)).max() <= tol


@pytest.mark.parametrize("name", ["g2_b8_d16_s1", "g4_b16_d32_s3_float32", "g5_b1_d32", "g5_tau002_b32_d64"])
def test_bf16_fast_kernels_match_bf16_model(name):
    m = IDX[name]
    v, t = golden_inputs(m)
    plan = nat.make_plan(m["B"], m["D"], 1, 0, nat.MODE_BF16)
    assert plan.fast_path == 1 and plan.Dpad == 128
    loss, gv, gt = run(v, t, m["temperature"], m["negative_weight"], "bf16")
    model = float(orc.bf16_operand_model_loss(v, t, m["temperature"], m["negative_weight"]))
    assert abs(loss.item() - model) <= 5e-5 * max(1.0, abs(model))
    arr = golden_arrays(name)
    scale = max(m["grad_v_absmax"], m["grad_t_absmax"])
    assert np.abs(gv.numpy() - arr["grad_v"]).max() <= 2e-2 * scale
    assert np.abs(gt.numpy() - arr["grad_t"]).max() <= 2e-2 * scale


def test_bf16_generic_kernels_match_bf16_model(monkeypatch):
    monkeypatch.setenv("CROSSCLR_DISABLE_FAST", "1")
    m = IDX["g4_b16_d32_s3_float32"]
    v, t = golden_inputs(m)
    assert nat.make_plan(16, 32, 1, 0, nat.MODE_BF16).fast_path == 0
    loss, gv, gt = run(v, t, m["temperature"], m["negative_weight"], "bf16")
    model = float(orc.bf16_operand_model_loss(v, t, m["temperature"], m["negative_weight"]))
    assert abs(loss.item() - model) <= 5e-5 * max(1.0, abs(model))
    arr = golden_arrays("g4_b16_d32_s3_float32")
    assert np.abs(gv.numpy() - arr["grad_v"]).max() <= 2e-2 * m["grad_v_absmax"]


@pytest.mark.parametrize("w", [0.0, -0.5, 1.7])
def test_zero_negative_and_large_negative_weight(w):
    # the masks are applied to the SCALED logit (-inf), so the sign / zero of negative_weight must not matter
    v, t = orc.make_inputs("randn", 24, 24, 23)
    ref = orc.streaming_loss_and_grads(v, t, 0.1, w)
    for mode, ltol, gtol in (("fp32", 1e-5, 2e-4), ("bf16", 3e-3, 2e-2)):
        loss, gv, gt = run(v, t, 0.1, w, mode)
        assert abs(loss.item() - float(ref["loss"])) <= ltol * max(1.0, abs(float(ref["loss"])))
        scale = max(ref["grad_v"].abs().max().item(), ref["grad_t"].abs().max().item())
        assert (gv.double() - ref["grad_v"]).abs().max().item() <= gtol * scale
        assert (gt.double() - ref["grad_t"]).abs().max().item() <= gtol * scale


@pytest.mark.parametrize("B,D", [(150, 32), (300, 40)])
def test_symmetric_forward_upper_triangle(B, D, monkeypatch):
    """bpad >= 256 -> several 256-row blocks: the fast forward evaluates only column tiles at/right of
    the diagonal block and recovers the mirrored tiles from column sums.  (300: bpad = 384, so one row
    block straddles the video/text boundary.)  Must agree with the full evaluation bit-for-bit-ish."""
    v, t = orc.make_inputs("randn", B, D, 5)
    sym = crossclr_amd.crossclr_loss(v, t, 0.05, 0.8, compute_mode="bf16").item()
    model = float(orc.bf16_operand_model_loss(v, t, 0.05, 0.8))
    assert abs(sym - model) <= 2e-6 * max(1.0, abs(model))
    monkeypatch.setenv("CROSSCLR_DISABLE_SYMMETRIC", "1")
    full = crossclr_amd.crossclr_loss(v, t, 0.05, 0.8, compute_mode="bf16").item()
    assert abs(sym - full) <= 1e-6 * max(1.0, abs(full))


@pytest.mark.parametrize("B,D", [(8, 16), (70, 48), (100, 70)])
def test_bf16_16row_backward_matches_reference(B, D, monkeypatch):
    """The 16-row-wavefront backward (v_mfma_f32_16x16x32_bf16; the default for 512 < D <= 1024) forced onto
    small widths so the emulator can run it."""
    monkeypatch.setenv("CROSSCLR_BWD_KERNEL", "16")
    assert nat.make_plan(B, D, 1, 0, nat.MODE_BF16).fast_bwd == 2
    v, t = orc.make_inputs("randn", B, D, 3)
    ref = orc.streaming_loss_and_grads(v, t, 0.05, 0.8)
    loss, gv, gt = run(v, t, 0.05, 0.8, "bf16")
    scale = max(ref["grad_v"].abs().max().item(), ref["grad_t"].abs().max().item())
    assert (gv.double() - ref["grad_v"]).abs().max().item() <= 2e-2 * scale
    assert (gt.double() - ref["grad_t"]).abs().max().item() <= 2e-2 * scale


def test_ragged_batch_crossing_a_tile_boundary():
    # B = 70: one full 64-column tile + a ragged one in the fast path; a ragged 128 tile in the generic path
    v, t = orc.make_inputs("randn", 70, 24, 17)
    ref = orc.streaming_loss_and_grads(v, t, 0.07, 0.6)
    for mode, ltol, gtol in (("fp32", 1e-5, 2e-4), ("bf16", 3e-3, 2e-2)):
        loss, gv, gt = run(v, t, 0.07, 0.6, mode)
        assert abs(loss.item() - float(ref["loss"])) <= ltol
        scale = ref["grad_v"].abs().max().item()
        assert (gv.double() - ref["grad_v"]).abs().max().item() <= gtol * scale
        assert (gt.double() - ref["grad_t"]).abs().max().item() <= gtol * scale


def test_grad_output_scaling_and_noncontiguous_rows():
    v, t = orc.make_inputs("randn", 12, 20, 5)
    _, g1, _ = run(v, t, 0.05, 0.8, "fp32", 1.0)
    _, g2, _ = run(v, t, 0.05, 0.8, "fp32", -2.5)
    assert torch.allclose(g2, -2.5 * g1, rtol=1e-5, atol=1e-12)
    wide = torch.zeros(12, 40)
    wide[:, ::2] = v
    vs = wide[:, ::2]  # column stride 2 -> host makes it row-major
    l1 = crossclr_amd.crossclr_loss(vs, t, 0.05, 0.8, compute_mode="fp32")
    l0 = crossclr_amd.crossclr_loss(v, t, 0.05, 0.8, compute_mode="fp32")
    assert l1.item() == l0.item()
    padded = torch.zeros(12, 32)
    padded[:, :20] = v
    lp = crossclr_amd.crossclr_loss(padded[:, :20], t, 0.05, 0.8, compute_mode="fp32")  # row stride 32, unit col stride
    assert lp.item() == l0.item()


def test_small_temperatures_take_the_two_pass_soft_max():
    """max |logit| > 128 (tau < 0.0078 at |w| <= 1): the reference still works (float64 soft-max with a row maximum,
    loss.py:60); so does the module, through crossclr_forward_rowmax + the _s entry points.  The plain C-ABI entry points
    keep refusing that range loudly (they have one common shift)."""
    import ctypes
    v, t = orc.make_inputs("randn", 8, 16, 1)
    plan = nat.make_plan(8, 16, 1, 0, nat.MODE_FP32)
    lib = nat.library()
    assert lib.crossclr_needs_row_shift(0.005, 0.8) == 1 and lib.crossclr_needs_row_shift(0.03, 0.8) == 0
    assert lib.crossclr_needs_row_shift(0.03, 4.0) == 1          # the bound is max(1, |w|) / tau
    x = torch.zeros(plan.operand_bytes, dtype=torch.uint8)
    part = torch.zeros(plan.fwd_ws_floats)
    rc = lib.crossclr_forward(ctypes.byref(plan), x.data_ptr(), x.data_ptr(), 1, 0, -1, 0.005, 0.8, part.data_ptr(), 0, None)
    assert rc == nat.E_RANGE and b"two-pass" in lib.crossclr_last_error()
    for tau in (0.005, 0.002, 0.0005):
        ref = orc.streaming_loss_and_grads(v, t, tau, 0.8)
        for mode, ltol in (("fp32", 1e-4), ("bf16", 2e-2)):
            loss, gv, gt = run(v, t, tau, 0.8, mode)
            assert abs(loss.item() - float(ref["loss"])) <= ltol * max(1.0, float(ref["loss"])), (tau, mode)
            if mode == "fp32":
                scale = max(ref["grad_v"].abs().max().item(), ref["grad_t"].abs().max().item())
                assert (gv.double() - ref["grad_v"]).abs().max().item() <= 1e-3 * scale
                assert (gt.double() - ref["grad_t"]).abs().max().item() <= 1e-3 * scale
    # small but inside the range: the common shift is active (1/tau = 100 > 64), result must still be right
    ref = orc.streaming_stats(v, t, 0.01, 0.8)
    out = crossclr_amd.crossclr_loss(v, t, temperature=0.01, negative_weight=0.8, compute_mode="fp32")
    assert abs(out.item() - float(ref["loss"])) <= 1e-4 * float(ref["loss"])


def test_small_temperature_aligned_pairs_give_the_reference_zero():
    # the positive pair dominates completely: the reference's float64 loss is exactly 0.0 with zero gradients
    m = IDX["g5_tau0002_aligned_b128_d96"]
    v, t = golden_inputs(m)
    loss, gv, gt = run(v, t, m["temperature"], m["negative_weight"], "fp32")
    # (logZ ~ 480 is carried in fp32 and the positive-pair logit is formed from an fp32 cosine: a 1e-5 cancellation floor,
    # a hundred times under the 1e-3 bar)
    assert m["loss"] == 0.0 and abs(loss.item()) <= 1e-4
    assert gv.abs().max().item() <= 1e-3 and gt.abs().max().item() <= 1e-3


def test_plan_geometry():
    p = nat.make_plan(8192, 512, 1, 0, nat.MODE_BF16)
    assert (p.bpad, p.Dpad, p.fast_path) == (8192, 512, 1)
    assert p.operand_bytes == 2 * 8192 * 512 * 2 and p.bwd_slices == 2 and p.gbuf_bytes == 2 * 2 * 8192 * 512 * 4 and p.fwd_blocks == 256
    p = nat.make_plan(100, 300, 8, 3, nat.MODE_FP32)
    assert (p.bpad, p.Dpad, p.fast_path, p.world, p.rank) == (128, 512, 0, 8, 3)
    p = nat.make_plan(100, 700, 1, 0, nat.MODE_BF16)
    assert (p.Dpad, p.fast_path, p.fast_bwd) == (768, 1, 2)   # 512 < D <= 1024: 4-wave forward, 16-row-wave backward
    p = nat.make_plan(100, 1500, 1, 0, nat.MODE_BF16)
    assert (p.Dpad, p.fast_path, p.fast_bwd) == (1536, 0, 0)  # wider: generic tiled forward ...
    assert p.stash_bytes > 0 and p.xf_bytes == p.operand_bytes   # ... that saves its exponentials for the D-slice backward (3 column parts, fragment-major copy available)
    assert [nat.make_plan(100, d, 1, 0, nat.MODE_BF16).Dpad for d in (1025, 1153, 2048, 2049, 3000, 4096)] == [1152, 1536, 2048, 2560, 3072, 4096]
    p = nat.make_plan(8192, 1536, 1, 0, nat.MODE_BF16)
    assert p.bwd_slices == 2 and p.gbuf_bytes == 2 * 2 * 8192 * 1536 * 4     # 128 row blocks x 3 parts x 2 slices = 3 rounds of 256
    assert nat.make_plan(8192, 2048, 1, 0, nat.MODE_BF16).bwd_slices == 1
    p = nat.make_plan(100, 5000, 1, 0, nat.MODE_BF16)
    assert (p.Dpad, p.fast_path) == (5120, 0) and p.stash_bytes > 0 and p.xf_bytes == p.operand_bytes      # round 6: 10 / 12 / 16 column parts up to D = 8192
    assert [nat.make_plan(100, d, 1, 0, nat.MODE_BF16).Dpad for d in (4097, 6000, 8192)] == [5120, 6144, 8192]
    p = nat.make_plan(100, 9000, 1, 0, nat.MODE_BF16)
    assert (p.fast_path, p.stash_bytes) == (0, 0)  # beyond 8192: the recomputing generic backward
    assert nat.make_plan(100, 512, 1, 0, nat.MODE_BF16).fast_bwd == 1 and nat.make_plan(100, 512, 1, 0, nat.MODE_FP32).fast_bwd == 0
    with pytest.raises(nat.CrossCLRNativeError):
        nat.make_plan(0, 16, 1, 0, nat.MODE_FP32)
    with pytest.raises(nat.CrossCLRNativeError):
        nat.make_plan(8, 16, 2, 2, nat.MODE_FP32)


def test_prenormalized_inputs_skip_the_normalisation_and_chain_through_autograd():
    """a producer that already emits unit rows calls the loss with prenormalized=True; F.normalize
    upstream (autograd) then supplies the normalise-backward, and values and gradients equal the plain path's."""
    v, t = orc.make_inputs("randn", 24, 40, 9)
    for mode, tol in (("fp32", 2e-6), ("bf16", 2e-6)):
        l0, gv0, gt0 = run(v, t, 0.05, 0.8, mode)
        vv = v.clone().requires_grad_(True)
        tt = t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(torch.nn.functional.normalize(vv, dim=1), torch.nn.functional.normalize(tt, dim=1),
                                          0.05, 0.8, compute_mode=mode, prenormalized=True)
        loss.backward()
        assert abs(loss.item() - l0.item()) <= tol * max(1.0, abs(l0.item()))
        scale = gv0.abs().max().item()
        assert (vv.grad - gv0).abs().max().item() <= 2e-5 * scale and (tt.grad - gt0).abs().max().item() <= 2e-5 * scale
    crit = crossclr_amd.CrossCLR_onlyIntraModality(0.05, 0.8, compute_mode="fp32", prenormalized=True)
    assert abs(crit(torch.nn.functional.normalize(v, dim=1), torch.nn.functional.normalize(t, dim=1)).item() -
               float(orc.streaming_stats(v, t, 0.05, 0.8)["loss"])) <= 1e-5


@pytest.mark.parametrize("B,D,weighted", [(70, 100, False), (40, 200, False), (70, 24, True), (33, 130, True),
                                          # several 128-row blocks: the saving forward evaluates the upper triangle only and stores
                                          # every off-diagonal fragment twice (as evaluated + transposed)
                                          (150, 24, False), (300, 40, True)])
def test_fp32_saved_exponentials_backward_equals_the_recomputing_one(B, D, weighted, monkeypatch):
    """compute_mode="fp32": the forward leaves its fp32 exponentials behind (plan.stash_bytes) and bwd_saved32_kernel forms
    the gradient product from them; CROSSCLR_DISABLE_SAVE=1 is the recomputing bwd_kernel.  Same weights, same products:
    agreement to fp32 rounding, for every slice width of D (64 / 128 / 256) and with per-sample weights."""
    assert nat.make_plan(B, D, 1, 0, nat.MODE_FP32).stash_bytes == (2 * nat.make_plan(B, D, 1, 0, nat.MODE_FP32).bpad) ** 2 * 4
    v, t = orc.make_inputs("randn", B, D, 11)
    kw = {}
    if weighted:
        g = torch.Generator().manual_seed(3)
        keep = lambda: (torch.rand(B, generator=g) > 0.3).float()
        kw = dict(negative_scale=(keep(), keep()), loss_weight=(torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5))

    def step():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="fp32", **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    ls, gvs, gts = step()
    monkeypatch.setenv("CROSSCLR_DISABLE_SAVE", "1")
    assert nat.make_plan(B, D, 1, 0, nat.MODE_FP32).stash_bytes == 0
    lr, gvr, gtr = step()
    assert abs(ls - lr) <= 1e-6 * max(1.0, abs(lr))
    scale = max(gvr.abs().max().item(), gtr.abs().max().item())
    assert (gvs - gvr).abs().max().item() <= 2e-6 * scale
    assert (gts - gtr).abs().max().item() <= 2e-6 * scale
    if not weighted:
        ref = orc.streaming_loss_and_grads(v, t, 0.05, 0.8)
        assert (gvs.double() - ref["grad_v"]).abs().max().item() <= 2e-5 * scale


@pytest.mark.parametrize("B,D,weighted", [(70, 600, False), (150, 1000, False), (40, 530, True)])
def test_wide_operands_take_the_saved_backward_in_two_column_parts(B, D, weighted, monkeypatch):
    """512 < D <= 1024 (bf16): the 4 x 32-row forward saves its exponentials (row blocks of 128: stash_tile_index with tpr = 4)
    and fast_bwd_saved_kernel<DK, SW, false, 2, 4> forms the gradient in two column parts of Dpad/2."""
    plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16)
    assert plan.fast_path == 1 and plan.Dpad in (768, 1024) and plan.stash_bytes > 0
    v, t = orc.make_inputs("randn", B, D, 23)
    kw = {}
    if weighted:
        g = torch.Generator().manual_seed(5)
        keep = lambda: (torch.rand(B, generator=g) > 0.3).float()
        kw = dict(negative_scale=(keep(), keep()), loss_weight=(torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5))

    def step():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="bf16", **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    ls, gvs, gts = step()
    monkeypatch.setenv("CROSSCLR_DISABLE_SAVE", "1")
    assert nat.make_plan(B, D, 1, 0, nat.MODE_BF16).stash_bytes == 0
    lr, gvr, gtr = step()
    assert abs(ls - lr) <= 1e-6 * max(1.0, abs(lr))
    scale = max(gvr.abs().max().item(), gtr.abs().max().item())
    assert (gvs - gvr).abs().max().item() <= 1e-2 * scale      # bf16 exponentials vs recomputed fp32 ones
    assert (gts - gtr).abs().max().item() <= 1e-2 * scale
    if not weighted:
        ref = orc.streaming_loss_and_grads(v, t, 0.05, 0.8)
        assert (gvs.double() - ref["grad_v"]).abs().max().item() <= 2e-2 * scale
        assert (gts.double() - ref["grad_t"]).abs().max().item() <= 2e-2 * scale


@pytest.mark.parametrize("B,D,weighted", [(150, 24, False), (300, 40, True), (150, 200, False), (140, 300, True)])
def test_saved_backward_with_mirrored_tiles(B, D, weighted, monkeypatch):
    """bpad >= 256: row blocks behind the first read column tiles the forward evaluated for ANOTHER row block -- stash tile
    (t, r32) holds E^T.  fast_bwd_dsl_kernel weighs them in their stored orientation and reads W^T through the transposing
    gather (bodies M->M, M->D, D->D; DK = 8 / 16 / 24 here).  Against the recomputing backward and the float64 oracle."""
    plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16)
    assert plan.fast_path == 1 and plan.Dpad <= 512 and plan.stash_bytes > 0 and plan.bpad >= 256
    v, t = orc.make_inputs("randn", B, D, 29)
    kw = {}
    if weighted:
        g = torch.Generator().manual_seed(5)
        keep = lambda: (torch.rand(B, generator=g) > 0.3).float()
        kw = dict(negative_scale=(keep(), keep()), loss_weight=(torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5))

    def step():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="bf16", **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    ls, gvs, gts = step()
    ls2, gvs2, gts2 = step()
    assert ls == ls2 and torch.equal(gvs, gvs2) and torch.equal(gts, gts2)
    monkeypatch.setenv("CROSSCLR_DISABLE_SAVE", "1")
    assert nat.make_plan(B, D, 1, 0, nat.MODE_BF16).stash_bytes == 0
    lr, gvr, gtr = step()
    assert abs(ls - lr) <= 1e-6 * max(1.0, abs(lr))
    scale = max(gvr.abs().max().item(), gtr.abs().max().item())
    assert (gvs - gvr).abs().max().item() <= 1e-2 * scale      # bf16 exponentials vs recomputed fp32 ones
    assert (gts - gtr).abs().max().item() <= 1e-2 * scale
    if not weighted:
        ref = orc.streaming_loss_and_grads(v, t, 0.05, 0.8)
        assert (gvs.double() - ref["grad_v"]).abs().max().item() <= 2e-2 * scale
        assert (gts.double() - ref["grad_t"]).abs().max().item() <= 2e-2 * scale


@pytest.mark.parametrize("B,D,weighted", [(150, 24, False), (300, 40, True), (150, 200, False), (140, 300, True), (130, 420, False),
                                          (130, 600, True)])
def test_fragment_major_backward_is_bit_identical_to_the_lds_staged_one(B, D, weighted, monkeypatch):
    """plan.xf_bytes > 0: crossclr_normalize_xf also writes the operand as MFMA B fragments; crossclr_backward_saved_xfp
    (fast_bwd_xfp_kernel: two tiles per barrier interval) and crossclr_backward_saved_xf (fast_bwd_dsl_kernel<..., XF>: one) load them
    straight into registers.  Every accumulator receives the same MFMA sequence as in the LDS-staged kernel, so all three gradients
    must agree BIT FOR BIT (DK = 8 / 16 / 24 / 32, two column parts at D = 600, mirrored and direct tiles, with and without sample weights)."""
    plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16)
    assert plan.fast_path == 1 and plan.stash_bytes > 0 and plan.xf_bytes == plan.operand_bytes and plan.bpad >= 256
    monkeypatch.setenv("CROSSCLR_XF_WIDTHS", "128,256,384,512,768,1024")     # (the module's default policy takes this path at 128 and 512 only)
    from crossclr_amd import loss as L      # (the step runs inside the library: its choice of backward is read from the reported layout)
    v, t = orc.make_inputs("randn", B, D, 31)
    kw = {}
    if weighted:
        g = torch.Generator().manual_seed(6)
        keep = lambda: (torch.rand(B, generator=g) > 0.3).float()
        kw = dict(negative_scale=(keep(), keep()), loss_weight=(torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5))

    def step():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="bf16", **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    lx, gvx, gtx = step()                                   # the pair kernel (two tiles per barrier interval): what a step runs
    assert L._last_step_backward_kernel == 3
    monkeypatch.setenv("CROSSCLR_XFP", "0")
    l1, gv1, gt1 = step()                                   # one tile per barrier interval
    assert L._last_step_backward_kernel == 2
    assert lx == l1 and torch.equal(gvx, gv1) and torch.equal(gtx, gt1)
    monkeypatch.delenv("CROSSCLR_XFP")
    monkeypatch.setenv("CROSSCLR_DISABLE_XF", "1")
    assert nat.make_plan(B, D, 1, 0, nat.MODE_BF16).xf_bytes == 0
    ll, gvl, gtl = step()
    assert L._last_step_backward_kernel == 1 and L._last_step_saved      # (this one went through crossclr_backward_saved)
    assert lx == ll and torch.equal(gvx, gvl) and torch.equal(gtx, gtl)
    monkeypatch.delenv("CROSSCLR_DISABLE_XF")
    # the XF entry point refuses a plan without the layout, and the prenormalized path (crossclr_pack_xf) agrees too
    vn, tn = torch.nn.functional.normalize(v, dim=1), torch.nn.functional.normalize(t, dim=1)

    def pstep():
        vv, tt = vn.clone().requires_grad_(True), tn.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="bf16", prenormalized=True, **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    a = pstep()
    monkeypatch.setenv("CROSSCLR_DISABLE_XF", "1")
    b_ = pstep()
    assert a[0] == b_[0] and torch.equal(a[1], b_[1]) and torch.equal(a[2], b_[2])


@pytest.mark.parametrize("B,D,weighted", [(150, 24, False), (300, 40, False), (150, 24, True), (70, 16, False)])
def test_generic_forward_only_evaluates_the_upper_triangle(B, D, weighted, monkeypatch):
    """compute_mode="fp32" under no_grad (BASELINE config 2's shape of call): fwd_sums_kernel<..., SYM> evaluates the column
    tiles at / right of the diagonal block only and recovers the mirrored tiles from column sums (workspace header kind 4).
    Same loss as the full evaluation (CROSSCLR_DISABLE_SYMMETRIC=1) and as the saving forward that a training step runs."""
    v, t = orc.make_inputs("randn", B, D, 19)
    kw = {}
    if weighted:
        g = torch.Generator().manual_seed(4)
        keep = lambda: (torch.rand(B, generator=g) > 0.3).float()
        kw = dict(negative_scale=(keep(), keep()), loss_weight=(torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5))
    with torch.no_grad():
        sym = crossclr_amd.crossclr_loss(v, t, 0.05, 0.8, compute_mode="fp32", **kw).item()
    train = crossclr_amd.crossclr_loss(v.clone().requires_grad_(True), t, 0.05, 0.8, compute_mode="fp32", **kw).item()
    monkeypatch.setenv("CROSSCLR_DISABLE_SYMMETRIC", "1")
    with torch.no_grad():
        full = crossclr_amd.crossclr_loss(v, t, 0.05, 0.8, compute_mode="fp32", **kw).item()
    assert abs(sym - full) <= 1e-6 * max(1.0, abs(full)) and abs(sym - train) <= 1e-6 * max(1.0, abs(train))
    if not weighted:
        ref = float(orc.streaming_loss_and_grads(v, t, 0.05, 0.8)["loss"])
        assert abs(sym - ref) <= 1e-5 * max(1.0, abs(ref))


@pytest.mark.parametrize("B,D,weighted,sym", [(70, 24, False, True), (150, 40, False, True), (150, 24, True, True), (150, 24, True, False)])
def test_two_pass_regime_saves_both_exponential_matrices(B, D, weighted, sym, monkeypatch):
    """tau = 0.004 (max |logit| 250 > 128), compute_mode="fp32": the second pass saves U[p][q] = exp2(x - shift_p) and
    Ut[p][q] = U[q][p] (crossclr_forward_save_s) and the backward forms U rz_p + Ut rz_q from them (crossclr_backward_saved_s);
    same results as the recomputing two-pass backward (CROSSCLR_DISABLE_SAVE=1), with the symmetric evaluation of the second pass
    and without it, and as the float64 oracle."""
    if not sym:
        monkeypatch.setenv("CROSSCLR_DISABLE_SYMMETRIC", "1")
    plan = nat.make_plan(B, D, 1, 0, nat.MODE_FP32)
    assert nat.library().crossclr_stash_bytes_s(ctypes.byref(plan)) == 2 * plan.stash_bytes > 0
    v, t = orc.make_inputs("randn", B, D, 29)
    kw = {}
    if weighted:
        g = torch.Generator().manual_seed(6)
        keep = lambda: (torch.rand(B, generator=g) > 0.3).float()
        kw = dict(negative_scale=(keep(), keep()), loss_weight=(torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5))

    def step():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.004, 0.8, compute_mode="fp32", **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    ls, gvs, gts = step()
    monkeypatch.setenv("CROSSCLR_DISABLE_SAVE", "1")
    lr, gvr, gtr = step()
    assert abs(ls - lr) <= 2e-6 * max(1.0, abs(lr))
    scale = max(gvr.abs().max().item(), gtr.abs().max().item())
    assert (gvs - gvr).abs().max().item() <= 1e-5 * scale and (gts - gtr).abs().max().item() <= 1e-5 * scale
    if not weighted:
        ref = orc.streaming_loss_and_grads(v, t, 0.004, 0.8)
        assert abs(ls - float(ref["loss"])) <= 1e-4 * max(1.0, abs(float(ref["loss"])))
        assert (gvs.double() - ref["grad_v"]).abs().max().item() <= 1e-3 * scale


def test_double_backward_matches_the_reference_and_never_returns_a_constant():
    """The reference's loss (trainer/loss.py:79-114) is twice differentiable.  The criterion carries that semantic (create_graph=True records the
    HIP backward as a node whose own backward is crossclr_second_order; the first-order path is untouched): a gradient penalty ||dL/dv||^2 differentiates to the same
    values as through the op-for-op oracle.  The ranking loss's closed-form backward is not twice differentiable and must raise."""
    v, t = orc.make_inputs("randn", 8, 16, 1)
    def penalty_grads(loss_fn):
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = loss_fn(vv, tt)
        gv, gt = torch.autograd.grad(loss, (vv, tt), create_graph=True)
        assert gv.requires_grad and gt.requires_grad
        pen = (gv.double() ** 2).sum() + 0.5 * (gt.double() ** 2).sum() + loss
        pen.backward()
        return loss.detach(), gv.detach(), vv.grad, tt.grad
    got = penalty_grads(lambda a, b: crossclr_amd.crossclr_loss(a, b, 0.05, 0.8, compute_mode="fp32"))
    want = penalty_grads(lambda a, b: orc.eager_loss(a, b, 0.05, 0.8))
    assert abs(got[0].item() - want[0].item()) <= 1e-6
    for g, w in zip(got[1:], want[1:]):
        assert (g.double() - w.double()).abs().max().item() <= 1e-5 * max(1.0, w.abs().max().item())
    # the ordinary backward of the same module still runs the kernels (no graph through it: grad mode is off inside)
    vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="fp32").backward()
    assert (vv.grad.double() - want[1].double()).abs().max().item() <= 1e-5 * want[1].abs().max().item()
    # per-sample weights go through the same closed form
    k = (torch.tensor([1.0, 0.0, 1.0, 0.5, 1.0, 1.0, 2.0, 1.0]), torch.ones(8))
    om = (torch.linspace(0.5, 1.5, 8), torch.ones(8))
    vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    lw = crossclr_amd.crossclr_loss(vv, tt, 0.05, 0.8, compute_mode="fp32", negative_scale=k, loss_weight=om)
    g1, = torch.autograd.grad(lw, vv, create_graph=True)
    vv2, tt2 = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
    crossclr_amd.crossclr_loss(vv2, tt2, 0.05, 0.8, compute_mode="fp32", negative_scale=k, loss_weight=om).backward()
    assert (g1.detach() - vv2.grad).abs().max().item() <= 1e-5 * vv2.grad.abs().max().item()
    a, b = torch.nn.functional.normalize(v, dim=1).requires_grad_(True), torch.nn.functional.normalize(t, dim=1).requires_grad_(True)
    mm = crossclr_amd.max_margin_loss(a, b, 0.1)
    with pytest.raises(RuntimeError, match="double backward"):
        torch.autograd.grad(mm, a, create_graph=True)
    mm2 = crossclr_amd.max_margin_loss(a, b, 0.1)
    mm2.backward()          # the ordinary backward is unaffected
    assert a.grad is not None


@pytest.mark.parametrize("B,D,weighted", [(40, 1100, False), (70, 1030, True), (24, 4700, False)])
def test_wide_bf16_plans_save_their_exponentials(B, D, weighted, monkeypatch):
    """1024 < D <= 8192, bf16: the generic symmetric forward leaves bf16 records in the register-resident layout (128-row blocks) and the
    D-slice saved backward runs as column parts of 384 / 512 columns -- against the streaming float64 oracle (reference: loss.py:83-112,
    shape-agnostic) and against the recomputing generic backward of the same library (CROSSCLR_DISABLE_SAVE=1)."""
    v, t = orc.make_inputs("randn", B, D, 77 + B)
    kw = {}
    if weighted:
        g = torch.Generator().manual_seed(5)
        kw = dict(negative_scale=(torch.rand(B, generator=g) + 0.5, (torch.rand(B, generator=g) > 0.2).float()),
                  loss_weight=(torch.rand(B, generator=g) + 0.5, torch.ones(B)))
    def step():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.03, 0.8, compute_mode="bf16", **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16)
    assert plan.stash_bytes > 0 and plan.fast_path == 0 and plan.xf_bytes == plan.operand_bytes
    ls, gvs, gts = step()
    # the same step with the column tiles taken as MFMA fragments from the fragment-major copy (pair kernel in 3 column parts; the
    # module's policy takes it from 4096 padded rows on): bit-identical to the LDS-staged kernel
    monkeypatch.setenv("CROSSCLR_XF_WIDTHS", str(plan.Dpad))
    from crossclr_amd import loss as L
    lx, gvx, gtx = step()
    assert L._last_step_backward_kernel == 3 and lx == ls and torch.equal(gvx, gvs) and torch.equal(gtx, gts)
    monkeypatch.delenv("CROSSCLR_XF_WIDTHS")
    monkeypatch.setenv("CROSSCLR_DISABLE_SAVE", "1")
    assert nat.make_plan(B, D, 1, 0, nat.MODE_BF16).stash_bytes == 0
    lr, gvr, gtr = step()
    assert abs(ls - lr) <= 1e-6 * max(1.0, abs(lr))
    scale = max(gvr.abs().max().item(), gtr.abs().max().item())
    # (the saved exponentials are rounded to bf16 once more when the weights are formed: a few 1e-3 of max|grad|)
    assert (gvs - gvr).abs().max().item() <= 1e-2 * scale and (gts - gtr).abs().max().item() <= 1e-2 * scale
    if not weighted:
        ref = orc.streaming_loss_and_grads(v, t, 0.03, 0.8)
        assert abs(ls - float(ref["loss"])) <= 1e-3
        assert (gvs.double() - ref["grad_v"]).abs().max().item() <= 1e-2 * scale
        assert (gts.double() - ref["grad_t"]).abs().max().item() <= 1e-2 * scale


@pytest.mark.parametrize("B,D,weighted", [(40, 1100, False), (70, 1030, True)])
def test_two_pass_regime_wide_bf16_plans_save_their_exponentials(B, D, weighted, monkeypatch):
    """tau = 0.004 on a wide bf16 plan (D > 1024: generic forward): there is no transposed launch of the D-slice kernel in column parts, so the
    second pass writes bf16 records of U AND of Ut[p][q] = exp2(x - shift_q) and the backward is two DIRECT launches (U with the rows'
    statistics, Ut with the columns') instead of the recomputing generic backward: same loss, gradients within the bf16 rounding."""
    plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16)
    n32 = 2 * plan.bpad // 32
    assert plan.fast_path == 0 and plan.Dpad == 1152
    assert nat.library().crossclr_stash_bytes_s(ctypes.byref(plan)) == 2 * n32 * n32 * 2048 + 2 * plan.bpad * 4
    v, t = orc.make_inputs("randn", B, D, 33)
    kw = {}
    if weighted:
        g = torch.Generator().manual_seed(9)
        keep = lambda: (torch.rand(B, generator=g) > 0.3).float()
        kw = dict(negative_scale=(keep(), keep()), loss_weight=(torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5))
    from crossclr_amd import loss as L

    def step():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.004, 0.8, compute_mode="bf16", **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    ls, gvs, gts = step()
    assert L._last_step_saved
    monkeypatch.setenv("CROSSCLR_DISABLE_SAVE", "1")
    lr, gvr, gtr = step()
    assert not L._last_step_saved          # (the recomputing path)
    assert abs(ls - lr) <= 1e-5 * max(1.0, abs(lr))
    scale = max(gvr.abs().max().item(), gtr.abs().max().item())
    assert (gvs - gvr).abs().max().item() <= 1e-2 * scale and (gts - gtr).abs().max().item() <= 1e-2 * scale
    if not weighted:
        ref = orc.streaming_loss_and_grads(v, t, 0.004, 0.8)
        assert abs(ls - float(ref["loss"])) <= 2e-2 * max(1.0, abs(float(ref["loss"])))
        assert (gvs.double() - ref["grad_v"]).abs().max().item() <= 3e-2 * scale


@pytest.mark.parametrize("B,D,weighted", [(70, 24, False), (150, 40, True), (40, 600, False)])
def test_two_pass_regime_bf16_plans_save_their_exponentials(B, D, weighted, monkeypatch):
    """tau = 0.004, compute_mode="bf16" (register-resident plans): the full second pass leaves bf16 records U[p][q] = exp2(x - shift_p) in
    the rectangular layout and the backward is TWO launches of the saved D-slice kernel -- direct (W = U rz_p) and transposed
    (W = U^T rz_q, accumulating) -- instead of the recomputing generic backward (CROSSCLR_DISABLE_SAVE=1): same loss, gradients within the
    bf16 rounding of the saved exponentials; D = 600: two column parts."""
    plan = nat.make_plan(B, D, 1, 0, nat.MODE_BF16)
    n32 = 2 * plan.bpad // 32
    assert plan.fast_path == 1 and nat.library().crossclr_stash_bytes_s(ctypes.byref(plan)) == n32 * n32 * 2048 + 2 * plan.bpad * 4
    v, t = orc.make_inputs("randn", B, D, 31)
    kw = {}
    if weighted:
        g = torch.Generator().manual_seed(8)
        keep = lambda: (torch.rand(B, generator=g) > 0.3).float()
        kw = dict(negative_scale=(keep(), keep()), loss_weight=(torch.rand(B, generator=g) + 0.5, torch.rand(B, generator=g) + 0.5))
    from crossclr_amd import loss as L

    def step():
        vv, tt = v.clone().requires_grad_(True), t.clone().requires_grad_(True)
        loss = crossclr_amd.crossclr_loss(vv, tt, 0.004, 0.8, compute_mode="bf16", **kw)
        loss.backward()
        return loss.item(), vv.grad, tt.grad
    ls, gvs, gts = step()
    assert L._last_step_saved
    monkeypatch.setenv("CROSSCLR_DISABLE_SAVE", "1")
    lr, gvr, gtr = step()
    assert not L._last_step_saved          # (the recomputing path)
    assert abs(ls - lr) <= 1e-5 * max(1.0, abs(lr))
    scale = max(gvr.abs().max().item(), gtr.abs().max().item())
    assert (gvs - gvr).abs().max().item() <= 1e-2 * scale and (gts - gtr).abs().max().item() <= 1e-2 * scale
    if not weighted:
        ref = orc.streaming_loss_and_grads(v, t, 0.004, 0.8)
        assert abs(ls - float(ref["loss"])) <= 2e-2 * max(1.0, abs(float(ref["loss"])))
        assert (gvs.double() - ref["grad_v"]).abs().max().item() <= 3e-2 * scale
