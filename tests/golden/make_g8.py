#!/usr/bin/env python3
"""Sampled-row goldens for BASELINE configs 4 and 5 at their REAL per-rank workload (8 ranks x b = 8192 rows, global
B = 65536): what rank 4 must produce for 64 of its rows.

The reference cannot run at this size (its [B, 2B] float64 temporaries would take ~1.6 TB), so these vectors come from the
oracle's streaming float64 form, which tests/test_oracle.py pins to the reference on every golden the reference CAN
produce.  They are therefore second-hand (said so in g8_index.json) -- the point is to exercise the sharded kernel path at
b = 8192 per rank against an independent float64 evaluation of exactly those rows.

    python tests/golden/make_g8.py            # ~10 min on 8 cores, 20 GB RAM
Inputs (the generator bench.py uses): rank r's rows = randn(8192, D) with seed 1234 + r, v drawn first then t.
config 4: D = 512, plain loss.  config 5: D = 1024, influential-sample weights from input-space features
(16 clusters + noise, seed 4321 + r, 256 features; threshold 0.9, temperature_weights 0.0035 -- bench.py --influential).
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import crossclr_oracle as orc          # noqa: E402
from oracle import influence_oracle as iorc        # noqa: E402

WORLD, b, RANK = 8, 8192, 4
TAU, W = 0.03, 0.8
CHUNKS = [(0, 16), (2731, 16), (5461, 16), (8176, 16)]      # 64 rows of rank 4: four runs of 16 (start inside the rank, length)


def global_inputs(D):
    vs, ts = [], []
    for r in range(WORLD):
        g = torch.Generator().manual_seed(1234 + r)
        vs.append(torch.randn(b, D, generator=g))
        ts.append(torch.randn(b, D, generator=g))
    return torch.cat(vs), torch.cat(ts)


def input_space_features():
    xs, ys = [], []
    for r in range(WORLD):
        g = torch.Generator().manual_seed(4321 + r)
        c = torch.randn(16, 256, generator=g)
        lab = torch.randint(0, 16, (b,), generator=g)
        xs.append(c[lab] + 0.1 * torch.randn(b, 256, generator=g))
        ys.append(c[lab] + 0.1 * torch.randn(b, 256, generator=g))
    return torch.cat(xs), torch.cat(ys)


def run(name, D, weighted):
    t0 = time.time()
    v, t = global_inputs(D)
    B = v.shape[0]
    if weighted:
        xv, xt = input_space_features()
        iw = iorc.influence_weights_streaming(xv, xt, 0.9, 0.0035)
        kv, kt, ov, ot = iw["keep_v"], iw["keep_t"], iw["omega_v"], iw["omega_t"]
    else:
        kv = kt = ov = ot = torch.ones(B, dtype=torch.float64)
    rows, lzv, lzt, gv, gt = [], [], [], [], []
    logz_all = None
    for c0, n in CHUNKS:
        lo = RANK * b + c0
        out = iorc.streaming_weighted_loss_and_grads(v, t, TAU, W, kv, kt, ov, ot, block=512, row_range=(lo, lo + n),
                                                     logz_all=logz_all)
        logz_all = (out["logZv_all"], out["logZt_all"])
        rows.append(np.arange(c0, c0 + n))
        lzv.append(out["logZv"].numpy()); lzt.append(out["logZt"].numpy())
        gv.append(out["grad_v"].numpy()); gt.append(out["grad_t"].numpy())
        print(f"{name}: rows {lo}..{lo + n} done, {time.time() - t0:.0f} s", flush=True)
    arrays = dict(rows_in_rank=np.concatenate(rows), logZv=np.concatenate(lzv), logZt=np.concatenate(lzt),
                  grad_v_rows=np.concatenate(gv), grad_t_rows=np.concatenate(gt))
    if weighted:
        arrays["pruned_fraction"] = np.array([1.0 - float(kv.mean()), 1.0 - float(kt.mean())])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    meta = dict(name=name, world=WORLD, rows_per_rank=b, rank=RANK, D=D, temperature=TAU, negative_weight=W, weighted=weighted,
                loss=float(out["loss"]), grad_absmax=float(max(np.abs(arrays["grad_v_rows"]).max(), np.abs(arrays["grad_t_rows"]).max())),
                source="oracle streaming float64 form (pinned to the reference by g1-g7); the reference cannot run at B = 65536",
                seconds=round(time.time() - t0))
    print(meta, flush=True)
    return meta


if __name__ == "__main__":
    which = sys.argv[1:] or ["c4", "c5"]
    path = os.path.join(HERE, "g8_index.json")
    idx = {m["name"]: m for m in json.load(open(path))["cases"]} if os.path.exists(path) else {}
    if "c4" in which:
        m = run("g8_c4_b65536_d512_rank4", 512, False); idx[m["name"]] = m
    if "c5" in which:
        m = run("g8_c5_b65536_d1024_rank4_weighted", 1024, True); idx[m["name"]] = m
    json.dump({"generator": "tests/golden/make_g8.py", "cases": sorted(idx.values(), key=lambda m: m["name"])}, open(path, "w"), indent=1)
