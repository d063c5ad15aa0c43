#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE.

Runs only in the build container (needs /root/reference).  The reference file
itself never leaves that container: what is committed is data -- generator kind
and seed, shapes, hyper-parameters, the SHA-256 of the inputs, and the
reference's outputs (loss, gradients or gradient samples).

The reference hard-codes `.cuda()` for its masks (trainer/loss.py:66,103,104);
in THIS process only, `torch.Tensor.cuda` is made the identity so the forward
runs on CPU tensors.  While at it the script proves the oracle's op-for-op form
bit-identical to the reference (torch.equal on loss and both gradients) for
every case it can afford to run twice, and refuses to write fixtures otherwise.

    python tests/golden/make_golden.py            # small + medium cases
    python tests/golden/make_golden.py --large    # adds B=4096 / B=8192 (12.5 GB RAM, ~1 min)
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

torch.Tensor.cuda = lambda self, *a, **k: self          # oracle process only
from trainer.loss import CrossCLR_onlyIntraModality as Reference  # noqa: E402
from oracle import crossclr_oracle as orc                          # noqa: E402

DT = {"float16": torch.float16, "bfloat16": torch.bfloat16, "float32": torch.float32,
      "float64": torch.float64}


def sha(*tensors):
    h = hashlib.sha256()
    for t in tensors:
        h.update(t.detach().contiguous().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()


def run_reference(v, t, tau, w):
    v = v.detach().clone().requires_grad_(True)
    t = t.detach().clone().requires_grad_(True)
    crit = Reference(temperature=tau, negative_weight=w)
    loss = crit(v, t)
    loss.backward()
    assert loss.dtype == torch.float64 and loss.dim() == 0
    return loss.detach(), v.grad, t.grad


def case(name, kind, B, D, seed, tau=0.03, w=0.8, dtype="float32", full=True, mutate=None,
         check_eager=True):
    v, t = orc.make_inputs(kind, B, D, seed, DT[dtype])
    if mutate == "zero_row":
        v[min(3, B - 1)] = 0
    loss, gv, gt = run_reference(v, t, tau, w)
    if check_eager:
        el, egv, egt = orc.eager_loss_and_grads(v, t, tau, w)
        ok = torch.equal(el, loss) and torch.equal(egv, gv) and torch.equal(egt, gt)
        if not ok:
            raise SystemExit(f"{name}: oracle eager form is NOT bit-identical to the reference")
    meta = dict(name=name, kind=kind, B=B, D=D, seed=seed, temperature=tau, negative_weight=w,
                dtype=dtype, mutate=mutate, input_sha256=sha(v, t), loss=float(loss),
                loss_repr=repr(float(loss)), grad_dtype=str(gv.dtype).replace("torch.", ""),
                grad_v_norm=float(gv.double().norm()), grad_t_norm=float(gt.double().norm()),
                grad_v_absmax=float(gv.double().abs().max()),
                grad_t_absmax=float(gt.double().abs().max()), eager_bit_identical=check_eager)
    arrays = {}
    if full:
        arrays["grad_v"] = gv.double().numpy() if dtype != "float32" else gv.numpy()
        arrays["grad_t"] = gt.double().numpy() if dtype != "float32" else gt.numpy()
    else:
        rows = np.linspace(0, B - 1, 8).astype(np.int64)
        arrays["rows"] = rows
        arrays["grad_v_rows"] = gv[rows].numpy()
        arrays["grad_t_rows"] = gt[rows].numpy()
    # per-row intermediates from the float64 closed form (what the kernels expose)
    if B <= 4096:
        st = orc.streaming_stats(v.float(), t.float(), tau, w)
        arrays["logZv"] = st["logZv"].numpy()
        arrays["logZt"] = st["logZt"].numpy()
        arrays["diag"] = st["diag"].numpy()
        meta["streaming_loss"] = float(st["loss"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    print(f"{name:28s} loss={meta['loss_repr']:>22s}  |gv|={meta['grad_v_norm']:.6e}")
    return meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true")
    args = ap.parse_args()
    torch.manual_seed(0)
    metas = []
    # G1: BASELINE config 1 (B=64, D=256)
    for s in (0, 7, 1234):
        metas.append(case(f"g1_b64_d256_s{s}", "randn", 64, 256, s))
    # G2: hand-checkable
    metas.append(case("g2_b8_d16_s1", "randn", 8, 16, 1))
    # G3
    metas.append(case("g3_b256_d512_s2", "randn", 256, 512, 2))
    # G4: dtype sweep
    for dt in ("float16", "bfloat16", "float32", "float64"):
        metas.append(case(f"g4_b16_d32_s3_{dt}", "randn", 16, 32, 3, dtype=dt))
    # G5: edge cases
    metas.append(case("g5_zero_row_b16_d32", "randn", 16, 32, 3, mutate="zero_row"))
    metas.append(case("g5_b1_d32", "randn", 1, 32, 3))
    metas.append(case("g5_w0_tau01_b16_d32", "randn", 16, 32, 3, tau=0.1, w=0.0))
    metas.append(case("g5_tau01_b16_d32", "randn", 16, 32, 3, tau=0.1))
    metas.append(case("g5_ragged_b100_d48", "randn", 100, 48, 5))
    metas.append(case("g5_ragged_b130_d200", "randn", 130, 200, 6, tau=0.05, w=0.5))
    metas.append(case("g5_tau002_b32_d64", "randn", 32, 64, 8, tau=0.02, w=1.0))
    # G6: aligned / clustered (bf16 stress), sampled rows only
    metas.append(case("g6_aligned_b2048_d512", "aligned", 2048, 512, 11, full=False))
    metas.append(case("g6_cluster_b2048_d512", "cluster", 2048, 512, 12, full=False))
    metas.append(case("g6_aligned_b256_d128", "aligned", 256, 128, 13))
    # G7: large
    if args.large:
        metas.append(case("g7_b4096_d512_s1234", "randn", 4096, 512, 1234, full=False))
        metas.append(case("g7_b8192_d512_s1234", "randn", 8192, 512, 1234, full=False,
                          check_eager=False))
    out = os.path.join(HERE, "index.json")
    prev = {}
    if os.path.exists(out):
        prev = {m["name"]: m for m in json.load(open(out))["cases"]}
    for m in metas:
        prev[m["name"]] = m
    json.dump({"generator": "tests/golden/make_golden.py",
               "reference": "amazon-science/crossmodal-contrastive-learning @ v1, trainer/loss.py",
               "torch": torch.__version__, "numpy": np.__version__,
               "cases": sorted(prev.values(), key=lambda m: m["name"])},
              open(out, "w"), indent=1)
    print("wrote", out)


if __name__ == "__main__":
    main()
