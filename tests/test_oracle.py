"""The oracle against the committed golden vectors (generated from the reference
by tests/golden/make_golden.py).  CPU only."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR, golden_arrays, golden_index, golden_inputs
from oracle import crossclr_oracle as orc

IDX = golden_index()
SMALL = [n for n, m in IDX.items() if m["B"] <= 256]
MEDIUM = [n for n, m in IDX.items() if 256 < m["B"] <= 2048]


@pytest.mark.parametrize("name", SMALL)
def test_eager_form_reproduces_reference_bits(name):
    m = IDX[name]
    v, t = golden_inputs(m)
    loss, gv, gt = orc.eager_loss_and_grads(v, t, m["temperature"], m["negative_weight"])
    assert loss.dtype == torch.float64 and loss.dim() == 0
    assert float(loss) == m["loss"], "eager oracle loss differs from the reference's bits"
    arr = golden_arrays(name)
    assert str(gv.dtype) == "torch." + m["grad_dtype"]
    if "grad_v" in arr:
        assert np.array_equal(gv.double().numpy(), arr["grad_v"].astype(np.float64))
        assert np.array_equal(gt.double().numpy(), arr["grad_t"].astype(np.float64))


@pytest.mark.parametrize("name", [n for n in SMALL if IDX[n]["dtype"] in ("float32", "float64")])
def test_streaming_form_matches_reference(name):
    m = IDX[name]
    v, t = golden_inputs(m)
    out = orc.streaming_loss_and_grads(v, t, m["temperature"], m["negative_weight"], block=37)
    assert abs(float(out["loss"]) - m["loss"]) <= 2e-6 * max(1.0, abs(m["loss"]))
    arr = golden_arrays(name)
    if "grad_v" in arr:
        for k, g in (("grad_v", out["grad_v"]), ("grad_t", out["grad_t"])):
            ref = arr[k].astype(np.float64)
            err = np.abs(g.numpy() - ref).max()
            # the reference forms its logits from an fp32 GEMM (loss.py:83-93): a 6e-8 rounding of a cosine is a
            # 6e-8/tau error of the logit, i.e. of the relative size of a soft-max weight
            rtol = max(5e-6, 2e-7 / m["temperature"])
            assert err <= rtol * max(np.abs(ref).max(), 1e-30), (k, err)
    # the stored per-row intermediates were produced from the fp32 cast of the inputs
    atol = 1e-9 if m["dtype"] == "float32" else 1e-5
    for k in ("logZv", "logZt", "diag"):
        assert np.allclose(out[k].numpy(), arr[k], rtol=0, atol=atol)


@pytest.mark.slow
@pytest.mark.parametrize("name", MEDIUM)
def test_streaming_form_medium_sampled_rows(name):
    m = IDX[name]
    v, t = golden_inputs(m)
    out = orc.streaming_loss_and_grads(v, t, m["temperature"], m["negative_weight"], block=512)
    assert abs(float(out["loss"]) - m["loss"]) <= 2e-6 * max(1.0, abs(m["loss"])) + 1e-12
    arr = golden_arrays(name)
    rows = arr["rows"]
    for k, g in (("grad_v_rows", out["grad_v"]), ("grad_t_rows", out["grad_t"])):
        ref = arr[k].astype(np.float64)
        err = np.abs(g[rows].numpy() - ref).max()
        # the reference's fp32 GEMM noise is ~3e-5 relative where gradients are ~1e-14 (aligned regime)
        assert err <= 2e-4 * max(m["grad_v_absmax"], m["grad_t_absmax"]), (k, err)


def test_sharded_equals_single_process():
    v, t = orc.make_inputs("randn", 48, 40, 21)
    full = orc.streaming_loss_and_grads(v, t, 0.05, 0.7, block=16)
    for world in (2, 3, 4):
        b = 48 // world
        for r in range(world):
            part = orc.sharded_loss_and_grads(v, t, world, r, 0.05, 0.7, block=16)
            assert float(part["loss"]) == pytest.approx(float(full["loss"]), abs=1e-12)
            assert torch.allclose(part["grad_v"], full["grad_v"][r * b:(r + 1) * b], atol=1e-12)
            assert torch.allclose(part["grad_t"], full["grad_t"][r * b:(r + 1) * b], atol=1e-12)


def test_w0_adds_B_to_each_denominator():
    # with w = 0 every intra entry contributes exp(0) = 1
    v, t = orc.make_inputs("randn", 12, 8, 4)
    st = orc.streaming_stats(v, t, 0.1, 0.0)
    vh = torch.nn.functional.normalize(v.double(), dim=1)
    th = torch.nn.functional.normalize(t.double(), dim=1)
    a = vh @ th.t() / 0.1
    assert torch.allclose(st["logZv"], torch.log(torch.exp(a).sum(1) + 12), atol=1e-6)


def test_bf16_operand_model_is_inside_the_loss_bar():
    # honesty check on the tolerance the GPU tests use for compute_mode="bf16"
    for name in ("g1_b64_d256_s0", "g1_b64_d256_s1234", "g3_b256_d512_s2"):
        m = IDX[name]
        v, t = golden_inputs(m)
        model = float(orc.bf16_operand_model_loss(v, t, m["temperature"], m["negative_weight"]))
        assert abs(model - m["loss"]) < 1e-3


@pytest.mark.skipif(not os.path.exists("/root/reference/trainer/loss.py"),
                    reason="the reference exists only in the build container")
def test_golden_recipe_regenerates_the_committed_fixtures_from_the_reference():
    """`make_golden.py --check` imports the reference BY FILE PATH, regenerates the small cases in memory and
    compares them with the committed index.json / *.npz bit for bit."""
    names = [n for n, m in IDX.items() if m["B"] <= 256]
    r = subprocess.run([sys.executable, os.path.join(GOLDEN_DIR, "make_golden.py"), "--check", "--only"] + names,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"golden check ok: {len(names)} cases" in r.stdout
