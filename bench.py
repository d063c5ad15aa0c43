#!/usr/bin/env python3
"""bench.py -- CrossCLR contrastive-loss hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 100 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one forward + backward of the loss over one synthetic batch that is already resident
in HBM: L2-normalisation, the fused similarity / soft-max-denominator kernel, the loss reduction,
the fused backward and the normalise-backward.  Workload = BASELINE.json configs[2]:
b = 8192 rows per GPU, D = 512, temperature 0.03, negative_weight 0.8, bf16 operands with fp32
accumulation; with N GPUs the global batch is B = 8192*N (configs[3] at N = 8), row-sharded,
with an RCCL all-gather of the packed normalised embeddings.

metric = contrastive pairs per second = B_global^2 / t_step (every video<->text pair of the global
batch is scored once per step); samples/s = B_global / t_step is reported next to it.
Rank 0 prints ONE JSON line.

Timing: `--prewarm` (default 40) untimed settle steps, then `--warmup` untimed steps, then exactly `--steps` timed steps
between barrier + synchronize fences, max over ranks -> `ms_per_step` / `value` (wall clock, the contract's number).  A second,
untimed pass of `--steps` steps is bracketed step by step by HIP events on the compute stream: `ms_per_step_event_median` is the
median of those (inside the timed window the two marker packets per step cost the device 5 us per step).  The settle steps exist because an MI355X that has just been handed to the process runs its first
~20-50 steps ~9 % slower (clock ramp); they are reported as `prewarm_steps` and named in `config.workload`.

`--gpus N` without a torch.distributed launcher (WORLD_SIZE unset) re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, so both
`python bench.py --gpus 8` and the driver's explicit torchrun command work.
`--fwd-only` (BASELINE configs[1] with `--mode fp32 --rows 4096`): forward under no_grad, metric "(fwd)".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

B_PER_GPU = 8192
DIM = 512
TAU = 0.03
NEG_W = 0.8
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
GOLDEN_LOSS_B8192_SEED1234 = 10.627098744839678  # tests/golden/index.json: g7_b8192_d512_s1234
GOLDEN_LOSS_B4096_SEED1234 = 9.919463972018582   # tests/golden/index.json: g7_b4096_d512_s1234


def csrc_sha():
    """Hash of the kernel sources: profiles/*_pmc.json records the value it was measured at (tools/make_pmc_json.py), so a
    traffic figure read from a committed profile can be flagged when the kernels have changed since."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "crossmodal-contrastive-learning_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".cpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def measured_traffic(b, d, mode, kernel_substr, kernel_suffix=None):
    """HBM bytes per launch of a kernel from the newest committed PMC summary under profiles/
    (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes).
    bench.py cannot run the profiler on itself; None when no summary matches this workload."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            j = json.load(open(path))
        except Exception:
            continue
        c = j.get("config", {})
        if (c.get("B"), c.get("D"), c.get("mode")) != (b, d, mode):
            continue
        for name, e in j.get("kernels", {}).items():
            if kernel_substr in name and (kernel_suffix is None or name.endswith(kernel_suffix)):
                best = {"bytes": e["hbm_bytes_corrected"], "source": os.path.relpath(path, ROOT), "csrc_sha": j.get("csrc_sha")}
    return best


def make_inputs(b, d, seed):
    """The synthetic workload (BASELINE.md section 3 / SURVEY.md 8(d)): v, t ~ N(0,1) fp32, v drawn first, then t, from one
    CPU generator.  Same stream as oracle.make_inputs("randn", ...) (checked by tests/test_bench_cpu.py) -- kept here so that
    the timed path does not import the oracle."""
    g = torch.Generator().manual_seed(seed)
    v = torch.randn(b, d, generator=g)
    t = torch.randn(b, d, generator=g)
    return v, t


def cpu_baseline(b, d, fwd_only=False):
    """The oracle's op-for-op restatement of the reference (bit-identical to it, see
    tests/golden/make_golden.py) timed on this box's host cores: bounded sample."""
    from oracle import crossclr_oracle as orc
    try:
        import psutil
        avail_gb = psutil.virtual_memory().available / 2**30
    except Exception:
        avail_gb = 0.0
    bb = b if avail_gb >= 24 else min(b, 4096)
    v, t = make_inputs(bb, d, 1234)
    orc.eager_loss_and_grads(v[:512], t[:512], TAU, NEG_W)  # warm the allocator / thread pool
    times = []
    budget_t0 = time.perf_counter()
    for i in range(3):
        t0 = time.perf_counter()
        if fwd_only:
            with torch.no_grad():
                loss = orc.eager_loss(v, t, TAU, NEG_W)
        else:
            loss, _, _ = orc.eager_loss_and_grads(v, t, TAU, NEG_W)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - budget_t0 > 25 and i >= 1:
            break
    rest = sorted(times[1:] or times)
    best = rest[(len(rest) - 1) // 2]      # median of the post-warm-up samples (the lower middle one of an even count)
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    # SURVEY.md 8(d) also asks for the reference's own CPU-runnable case (BASELINE config 1: B = 64, D = 256): median of 20 steps
    small = None
    if not fwd_only:
        vs, ts = make_inputs(64, 256, 1234)
        ts_ = []
        for i in range(21):
            t0 = time.perf_counter()
            ls, _, _ = orc.eager_loss_and_grads(vs, ts, TAU, NEG_W)
            ts_.append(time.perf_counter() - t0)
        ts_ = sorted(ts_[1:])
        small = {"B": 64, "D": 256, "seconds_per_step": ts_[len(ts_) // 2], "pairs_per_s": 64 * 64 / ts_[len(ts_) // 2], "loss": float(ls.detach())}
    return {"value": bb * bb / best, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model, "config1_b64_d256": small,
            "samples_per_s": bb / best, "seconds_per_step": best, "seconds_all_steps": [round(x, 4) for x in times],
            "host_cpus": os.cpu_count(),
            "loss": float(loss.detach()),
            "sample": f"oracle.eager_loss{'' if fwd_only else '_and_grads'} (op-for-op reference restatement) "
                      f"{'fwd' if fwd_only else 'fwd+bwd'}, fp32 inputs, "
                      f"B={bb} D={d}, {len(times)} steps (first = warm-up), median of the rest"}


def sustained_mfma(dev):
    """What a loop of nothing but v_mfma_f32_32x32x16_bf16 sustains on THIS device in THIS run (crossclr_mfma_sustained: 256 blocks x 4
    waves x 16 independent chains; pseudo-random operands = toggling data, and all-zero operands beside it): the package power
    limit, not the schedule, sets this rate.  HIP events, median of 5 launches of ~1.3 ms after 3 settle launches."""
    from crossclr_amd import _native as nat
    lib = nat.library()
    blocks, iters = 256, 4096
    out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    flop = blocks * 4.0 * iters * 16 * 32768
    res = {}
    for name, zero in (("random_operands", 0), ("zero_operands", 1)):
        for _ in range(3):
            nat.check(lib.crossclr_mfma_sustained(out.data_ptr(), blocks, iters, 12345, zero, stream))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for i, (a, z) in enumerate(ev):
            a.record()
            nat.check(lib.crossclr_mfma_sustained(out.data_ptr(), blocks, iters, 777 + i, zero, stream))
            z.record()
        torch.cuda.synchronize(dev)
        ms = sorted(a.elapsed_time(z) for a, z in ev)
        res[name] = {"tflops": round(flop / (ms[2] * 1e-3) / 1e12, 1), "ms_per_launch": round(ms[2], 4)}
    res["kernel"] = "mfma_sustained_kernel (crossclr_mfma_sustained): 256 blocks x 4 waves x 4096 x 16 MFMAs, 16 accumulator chains"
    return res


def library_gemm(dev, n2, d):
    """The same-box practical roof of the dominant kernel's product: torch.matmul (hipBLASLt / rocBLAS) of a bf16 [n2, n2] matrix with a
    bf16 [n2, d] one, fp32 accumulation -- G = W X of the saved backward (n2 = 2 b stacked rows, 8 b^2 D flop) WITHOUT forming W from the
    saved exponentials and the statistics.  Random operands (toggling data: the package power limit applies to the library too), median of
    20 launches after 5, HIP events."""
    g = torch.Generator(device="cpu").manual_seed(99)
    a = torch.randn(n2, n2, generator=g, dtype=torch.float32).to(dev).to(torch.bfloat16)
    x = torch.randn(n2, d, generator=g, dtype=torch.float32).to(dev).to(torch.bfloat16)
    out = torch.empty(n2, d, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        torch.matmul(a, x, out=out)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for s, e in ev:
        s.record()
        torch.matmul(a, x, out=out)
        e.record()
    torch.cuda.synchronize(dev)
    ms = sorted(s.elapsed_time(e) for s, e in ev)[10]
    flop = 2.0 * n2 * n2 * d
    del a, x, out
    return {"ms": round(ms, 4), "tflops": round(flop / (ms * 1e-3) / 1e12, 1),
            "what": f"torch.matmul bf16 [{n2},{n2}] @ [{n2},{d}] -> bf16, fp32 accumulate (the library's kernel for the dominant kernel's "
                    f"product, operands ready-made), median of 20"}


def sustained_series(step, dev, nsteps, chunk=100):
    """`nsteps` more steps of the same workload, HIP events around every `chunk` steps: does the step time hold for a second or
    more on a part that runs at its package power limit?"""
    n = max(1, nsteps // chunk)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev[0].record()
    for c in range(n):
        for _ in range(chunk):
            step()
        ev[c + 1].record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    per = [ev[c].elapsed_time(ev[c + 1]) / chunk for c in range(n)]
    return {"steps": n * chunk, "ms_per_step": round(wall * 1e3 / (n * chunk), 5), "first_100": round(per[0], 5), "last_100": round(per[-1], 5),
            "min_chunk": round(min(per), 5), "max_chunk": round(max(per), 5), "chunk_steps": chunk,
            "note": "wall clock over all steps; first/last/min/max = HIP-event time per step of a 100-step chunk"}


def secondary_lines(dev):
    """Short driver-timed measurements of the other configurations (12 warm + 12 timed steps each, ~1 s together): HIP-event median of
    the step, loss against the reference golden where one exists, algorithmic rate of the dominant kernel."""
    import crossclr_amd
    from crossclr_amd import _profile
    out = {}

    def timed(fn, n=12, warm=12):      # (warm: plan, allocator and the one-off self-test of a new kernel instantiation settle in the first steps)
        for _ in range(warm):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, z in ev:
            a.record()
            last = fn()
            z.record()
        torch.cuda.synchronize(dev)
        ms = sorted(a.elapsed_time(z) for a, z in ev)
        return ms[len(ms) // 2], last

    def case(name, rows, dim, mode, fwd_only, influential, golden):
        v, t = make_inputs(rows, dim, 1234)
        v, t = v.to(dev).requires_grad_(not fwd_only), t.to(dev).requires_grad_(not fwd_only)
        extra = ()
        if influential:
            g = torch.Generator().manual_seed(4321)
            c = torch.randn(16, 256, generator=g)
            lab = torch.randint(0, 16, (rows,), generator=g)
            extra = ((c[lab] + 0.1 * torch.randn(rows, 256, generator=g)).to(dev), (c[lab] + 0.1 * torch.randn(rows, 256, generator=g)).to(dev))
            crit = crossclr_amd.CrossCLR(TAU, 0.0035, NEG_W, 0.9, compute_mode=mode).to(dev)
        else:
            crit = crossclr_amd.CrossCLR_onlyIntraModality(TAU, NEG_W, compute_mode=mode).to(dev)

        def step():
            if fwd_only:
                with torch.no_grad():
                    return crit(v, t, *extra)
            v.grad = t.grad = None
            loss = crit(v, t, *extra)
            loss.backward()
            return loss
        ms, loss = timed(step)
        sw = crossclr_amd.influential_sample_weights(extra[0], extra[1], 0.9, 0.0035) if influential else (None, None)
        st = _profile.stage_times(v.detach(), t.detach(), TAU, NEG_W, mode, iters=7, warmup=2, negative_scale=sw[0], loss_weight=sw[1])   # (median of 7: 3 gave outliers)
        peak = PEAK_BF16_TFLOPS if mode == "bf16" else PEAK_F32_TFLOPS
        dom = "forward" if fwd_only else "step_backward"      # (under no_grad nothing is saved: crossclr_forward, not crossclr_forward_save)
        flops = (6.0 if fwd_only else 8.0) * rows * rows * dim
        tf = flops / (st[dom] * 1e-3) / 1e12
        ex_frac = None
        if fwd_only:   # (symmetric forward: 4.06 b^2 D executed for 6 b^2 D algorithmic -- the algorithmic fraction can pass 1)
            ex_frac = round(4.0 * rows * rows * dim * (1.0 + (256.0 if st["fast_path"] else 128.0) / (2.0 * rows)) / (st[dom] * 1e-3) / 1e12 / peak, 4)
        out[name] = {"ms_per_step_event_median": round(ms, 4), "pairs_per_s": rows * rows / (ms * 1e-3), "loss": float(loss.detach()),
                     "dominant_kernel_frac_executed": ex_frac,
                     "loss_delta_vs_reference": abs(float(loss.detach()) - golden) if golden is not None else None,
                     "dominant_kernel": ("forward" if fwd_only else "backward") + (" (saved exponentials)" if st.get("saved_path") and not fwd_only else ""),
                     "dominant_kernel_ms": round(st[dom], 4), "dominant_kernel_algorithmic_tflops": round(tf, 2),
                     "dominant_kernel_frac_of_peak": round(tf / peak, 4), "peak_tflops": peak,
                     "workload": f"b={rows} D={dim} {mode} {'fwd' if fwd_only else 'fwd+bwd'}" + (" + influential-sample weights" if influential else "")}
    # BASELINE config 1 (the reference's own CPU-runnable case, B = 64, D = 256, exact fp32): a launch-bound step -- what it costs is the host
    # path (five launches + autograd), see cpu_baseline.config1_b64_d256 for the same step through the reference's ops on this box's CPU
    v1, t1 = make_inputs(64, 256, 1234)
    v1, t1 = v1.to(dev).requires_grad_(True), t1.to(dev).requires_grad_(True)
    crit1 = crossclr_amd.CrossCLR_onlyIntraModality(TAU, NEG_W, compute_mode="fp32").to(dev)

    def step1():
        v1.grad = t1.grad = None
        l = crit1(v1, t1)
        l.backward()
        return l
    for _ in range(50):
        step1()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(200):
        l1 = step1()
    torch.cuda.synchronize(dev)
    out["config1_fp32_b64_d256_fwd_bwd"] = {"ms_per_step_wall": round((time.perf_counter() - t0) / 200 * 1e3, 4), "loss": float(l1.detach()),
                                             "workload": "b=64 D=256 fp32 fwd+bwd, wall over 200 steps (host-bound: module call + autograd + five launches)"}
    case("fp32_b8192_fwd_bwd", 8192, 512, "fp32", False, False, GOLDEN_LOSS_B8192_SEED1234)
    case("config2_fp32_fwd_b4096", 4096, 512, "fp32", True, False, GOLDEN_LOSS_B4096_SEED1234)
    case("d1024_bf16_fwd_bwd", 8192, 1024, "bf16", False, False, None)
    case("d1024_influential", 8192, 1024, "bf16", False, True, None)
    case("d1536_bf16_fwd_bwd", 8192, 1536, "bf16", False, False, None)     # beyond the register-resident forward: generic forward + saved D-slice backward in 3 column parts
    # exact-fp32 sharded run in the two-pass regime (tau = 0.005), rank 0 of 2 through the C-ABI: the block against the other rank from saved
    # exponentials (U and Ut) -- the last shape whose remote blocks recomputed the similarity product
    v, t = make_inputs(2 * 8192, 512, 1234)
    rb = _profile.remote_block_times(v.to(dev), t.to(dev), 0.005, NEG_W, iters=5, warmup=2, recompute=True)
    tf = 8.0 * 8192 * 8192 * 512 / (rb["backward_rect_saved"] * 1e-3) / 1e12
    out["tau0005_fp32_sharded_block"] = {
        "forward_rect_save_ms": round(rb["forward_rect_save"], 4), "backward_rect_saved_ms": round(rb["backward_rect_saved"], 4),
        "forward_recompute_path_ms": round(rb["forward_recompute_path"], 4), "backward_recompute_ms": round(rb["backward_recompute"], 4),
        "stash_gib": round(rb["stash_bytes"] / 2.0 ** 30, 3), "dominant_kernel": "backward of the remote block (saved exponentials U and Ut)",
        "dominant_kernel_ms": round(rb["backward_rect_saved"], 4), "dominant_kernel_algorithmic_tflops": round(tf, 2),
        "dominant_kernel_frac_of_peak": round(tf / PEAK_F32_TFLOPS, 4), "peak_tflops": PEAK_F32_TFLOPS,
        "workload": "rank 0 of 2 through the C-ABI on one GPU: b=8192 rows/rank, D=512, fp32, tau=0.005 (two-pass soft-max), the 16384 x 16384 block "
                    "against the other rank: second forward pass that saves + saved backward; median of 5 launches"}
    # wide bf16 plan (D = 1536) in a sharded run, rank 0 of 2 through the C-ABI: the block against the other rank from saved bf16 records
    v, t = make_inputs(2 * 8192, 1536, 1234)
    from crossclr_amd import _native as nat
    rb = _profile.remote_block_times(v.to(dev), t.to(dev), TAU, NEG_W, iters=5, warmup=2, recompute=True, mode=nat.MODE_BF16)
    tf = 8.0 * 8192 * 8192 * 1536 / (rb["backward_rect_saved"] * 1e-3) / 1e12
    out["d1536_bf16_sharded_block"] = {
        "forward_rect_save_ms": round(rb["forward_rect_save"], 4), "backward_rect_saved_ms": round(rb["backward_rect_saved"], 4),
        "forward_recompute_path_ms": round(rb["forward_recompute_path"], 4), "backward_recompute_ms": round(rb["backward_recompute"], 4),
        "stash_gib": round(rb["stash_bytes"] / 2.0 ** 30, 3), "dominant_kernel": "backward of the remote block (saved bf16 records, D-slice kernel in 3 column parts)",
        "dominant_kernel_ms": round(rb["backward_rect_saved"], 4), "dominant_kernel_algorithmic_tflops": round(tf, 2),
        "dominant_kernel_frac_of_peak": round(tf / PEAK_BF16_TFLOPS, 4), "peak_tflops": PEAK_BF16_TFLOPS,
        "workload": "rank 0 of 2 through the C-ABI on one GPU: b=8192 rows/rank, D=1536, bf16, tau=0.03, the 16384 x 16384 block against the other "
                    "rank: generic forward that saves + saved backward; median of 5 launches"}
    # round 6: the double backward (create_graph=True) at the headline shape -- forward + backward + crossclr_second_order (closed form on the
    # device, exact fp32) behind a bf16 step; what the reference does with 12 GB of float64 [B, 2B] tensors per autograd level
    v, t = make_inputs(8192, 512, 1234)
    v, t = v.to(dev).requires_grad_(True), t.to(dev).requires_grad_(True)
    crit = crossclr_amd.CrossCLR_onlyIntraModality(TAU, NEG_W, compute_mode="bf16").to(dev)

    def penalty_step():
        gv, gt = torch.autograd.grad(crit(v, t), (v, t), create_graph=True)
        return torch.autograd.grad((gv.double() ** 2).sum() + (gt.double() ** 2).sum(), (v, t))[0]
    torch.cuda.reset_peak_memory_stats(dev)
    base = torch.cuda.memory_allocated(dev)
    ms, pv = timed(penalty_step, n=5, warm=2)
    out["second_order_b8192_gradient_penalty"] = {
        "ms_per_step_event_median": round(ms, 3), "peak_memory_mib_above_inputs": round((torch.cuda.max_memory_allocated(dev) - base) / 2 ** 20, 1),
        "finite": bool(torch.isfinite(pv).all()),
        "workload": "b=8192 D=512: loss -> autograd.grad(create_graph=True) -> gradient of ||dL/dv||^2 + ||dL/dt||^2 (bf16 first-order step, exact-fp32 "
                    "closed-form double backward: crossclr_second_order; (3 + 3 + 2) tiled products of 2 (2b)^2 D + the fp32 first-order pieces)"}
    # round 6: the reference's other loss (MaxMargin_coot, loss.py:17-41), exact fp32, forward + backward from the saved hinge mask
    im, s = make_inputs(8192, 512, 77)
    im = torch.nn.functional.normalize(im, dim=1).to(dev).requires_grad_(True)
    s = torch.nn.functional.normalize(s, dim=1).to(dev).requires_grad_(True)

    def mm_step():
        im.grad = s.grad = None
        loss = crossclr_amd.max_margin_loss(im, s, 0.1, compute_mode="fp32")
        loss.backward()
        return loss
    ms, loss = timed(mm_step, n=7, warm=3)
    out["maxmargin_fp32_b8192_fwd_bwd"] = {"ms_per_step_event_median": round(ms, 4), "loss": float(loss.detach()),
                                            "workload": "MaxMargin_coot(margin=0.1) b=8192 D=512 exact fp32 fwd+bwd (one score pass for both directions that "
                                                        "also saves the hinge mask; backward = one product with the mask)"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--mode", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--rows", type=int, default=B_PER_GPU, help="rows per GPU (default = BASELINE config)")
    ap.add_argument("--dim", type=int, default=DIM)
    ap.add_argument("--prewarm", type=int, default=40,
                    help="untimed device settle steps BEFORE the --warmup steps (GPU clocks / allocator reach steady state "
                         "only after ~20-50 steps: 0.79 -> 0.72 ms/step); reported in the JSON line")
    ap.add_argument("--settle-cap", type=int, default=400,
                    help="after --prewarm: further untimed 20-step chunks until three consecutive ones agree within 1 %% (at most this many "
                         "steps; 0 = none); reported as settle_steps_used")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the short secondary measurements (fp32, config 2, D = 1024)")
    ap.add_argument("--sustained-steps", type=int, default=2000,
                    help="extra untimed-by-the-contract steps AFTER the timed region whose time series (per 100 steps) is reported "
                         "as `sustained` (about one second at the default workload); 0 = skip")
    ap.add_argument("--fwd-only", action="store_true", help="forward under no_grad only (BASELINE configs[1])")
    ap.add_argument("--selftest-emu", action="store_true",
                    help="TESTS ONLY (tests/test_bench_cpu.py): CPU tensors, gloo, the host-emulation build of the kernels -- "
                         "exercises the multi-rank plumbing, the timing fences and the JSON line without a GPU")
    ap.add_argument("--exchange-deadline", type=float, default=None,
                    help="N >= 3: seconds each operand-exchange form after the first may take before the run is ended with the line of "
                         "the forms that finished (default 120; 0 = no watchdog, the default on the host emulation)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TESTS ONLY (tests/test_gpu_ranks_share_gpu.py): all N ranks on cuda:0, collectives over gloo on device tensors "
                         "(RCCL refuses two ranks on one device) -- the sharded host path, the three operand exchanges and this file's "
                         "multi-rank diagnostics with the REAL kernels in separate processes on a one-GPU box; not a measurement")
    ap.add_argument("--influential", action="store_true",
                    help="BASELINE config 5: influential-sample pruning + loss weighting from synthetic input-space "
                         "features (crossclr_amd.CrossCLR); not the default workload")
    args = ap.parse_args()

    # RCCL prints a version banner on the C-level stdout of every rank.  Keep this process's stdout to
    # exactly ONE JSON line: everything else that lands on fd 1 is sent to stderr; the result goes to
    # the saved descriptor.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become one (one rank per GPU, rendezvous on the loopback address)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)

    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    import crossclr_amd
    from crossclr_amd import _native as nat
    from crossclr_amd import _profile

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    emu = args.selftest_emu
    if args.exchange_deadline is None:
        args.exchange_deadline = 0.0 if emu else 120.0
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu import build_emu
        nat.use_library_for_testing(build_emu.build())
        dev = torch.device("cpu")
    else:
        assert nat.backend() == "hip-gfx950", "bench needs the HIP library"
        torch.cuda.set_device(0 if args.share_gpu else local_rank)
        dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emu or args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        group = dist.group.WORLD

    b, d = args.rows, args.dim
    v, t = make_inputs(b, d, 1234 + rank)
    v = v.to(dev).requires_grad_(not args.fwd_only)
    t = t.to(dev).requires_grad_(not args.fwd_only)
    if args.influential:
        # input-space features: 16 clusters + noise (so that connectivities differ and the threshold prunes a part)
        g = torch.Generator().manual_seed(4321 + rank)
        c = torch.randn(16, 256, generator=g)
        lab = torch.randint(0, 16, (b,), generator=g)
        xv = (c[lab] + 0.1 * torch.randn(b, 256, generator=g)).to(dev)
        xt = (c[lab] + 0.1 * torch.randn(b, 256, generator=g)).to(dev)
        crit = crossclr_amd.CrossCLR(TAU, 0.0035, NEG_W, 0.9, compute_mode=args.mode, process_group=group).to(dev)
    else:
        crit = crossclr_amd.CrossCLR_onlyIntraModality(TAU, NEG_W, compute_mode=args.mode, process_group=group).to(dev)

    def step():
        if args.fwd_only:
            with torch.no_grad():
                return crit(v, t, xv, xt) if args.influential else crit(v, t)
        v.grad = None
        t.grad = None
        loss = crit(v, t, xv, xt) if args.influential else crit(v, t)   # the O(B D) weight recipe is inside the step
        loss.backward()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        if not emu:
            torch.cuda.synchronize(dev)

    from crossclr_amd import loss as L

    def timed_run():
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize fences (max over ranks), HIP events around
        every step; with several ranks a few extra diagnostic steps with events around every wait on a collective."""
        for _ in range(args.warmup):
            loss = step()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = step()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = tmax.item()
        # HIP events around every step of a SECOND pass of K steps, outside the timed region: an event record is a marker packet in the
        # stream, two per step cost the device 5 us per step (profiles/r05p_window.txt) -- the contract's window holds the steps alone
        ev = None if emu else [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        if ev:
            for i in range(args.steps):
                ev[i][0].record()
                loss = step()
                ev[i][1].record()
            fence()
        ev_ms = sorted(a.elapsed_time(z) for a, z in ev) if ev else []
        # ---- multi-rank diagnostics (untimed, after the measured region): per rank, the HIP-event time of a step and how much of it
        # the compute stream spent WAITING for collectives (events around every wait: communication not hidden behind compute) ----
        per_rank = None
        if world > 1:
            diag_steps = 0 if emu else 5
            waits, steps_ms = [], []
            for _ in range(diag_steps):
                L._comm_trace = []
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                step()
                e1.record()
                torch.cuda.synchronize(dev)
                steps_ms.append(e0.elapsed_time(e1))
                by_tag = {}
                for tag, a, z in L._comm_trace:
                    by_tag[tag] = by_tag.get(tag, 0.0) + a.elapsed_time(z)
                waits.append(by_tag)
            L._comm_trace = None
            mine = {"rank": rank, "exchange": L._last_exchange_mode,
                    "step_ms_event_median": sorted(steps_ms)[len(steps_ms) // 2] if steps_ms else None,
                    "timed_step_ms_event_median": ev_ms[len(ev_ms) // 2] if ev_ms else None}
            if waits:
                tags = sorted({k for w in waits for k in w})
                med = lambda xs: sorted(xs)[len(xs) // 2]
                mine["exposed_comm_ms_by_wait"] = {k: round(med([w.get(k, 0.0) for w in waits]), 4) for k in tags}
                mine["exposed_comm_ms"] = round(med([sum(w.values()) for w in waits]), 4)
                mine["compute_ms"] = round(mine["step_ms_event_median"] - mine["exposed_comm_ms"], 4)
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            per_rank = gathered
        return {"t_step": elapsed / args.steps, "loss": float(loss.item()), "ev_ms": ev_ms, "per_rank": per_rank}

    for _ in range(max(0, args.prewarm)):
        step()
    # The timed window starts only once the device has settled: chunks of 20 steps (each between synchronize fences, like the timed
    # region itself) until three consecutive chunks agree within 1 % -- at most `--settle-cap` steps (a freshly leased MI355X ramps its
    # clocks for 50-200 steps; a 20-step window taken before that reads 5-7 % slow).  Single-rank runs only: every rank would have to take
    # the same decision.  Reported as `settle_steps_used`.
    settle_used, settle_chunks = 0, []
    if world == 1 and not emu and args.settle_cap > 0:
        while settle_used < args.settle_cap:
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize(dev)
            settle_chunks.append((time.perf_counter() - t0) / 20 * 1e3)
            settle_used += 20
            last = settle_chunks[-3:]
            if len(last) == 3 and max(last) <= 1.01 * min(last):
                break
    # From 3 ranks on the packed operands can travel three ways (loss._OperandExchange); which one wins is a property of the node, so the
    # timed region is run once per form, back to back (each with its own warm-up), and the line reports all three: `value` / `ms_per_step`
    # are the fastest form's K steps, `per_exchange` the table.  CROSSCLR_EXCHANGE pins one form (then only that one is run).
    per_exchange = None
    if world >= 3 and args.mode == "bf16" and os.environ.get("CROSSCLR_EXCHANGE") is None and not args.fwd_only:
        per_exchange, runs = {}, {}
        # The first form (one standard all_gather_into_tensor) is the line's safety net: should a later form hang or raise on some rank
        # of a node this code has never seen (its peers would then sit in their collectives), every rank's watchdog ends the process
        # after `--exchange-deadline` seconds and rank 0 prints the line of the forms that did finish, the failure named in `per_exchange`.
        import threading
        state = {"trying": None}

        def give_up():
            if rank == 0:
                best = min(runs, key=lambda k: runs[k]["t_step"])
                r = runs[best]
                table = {k: {"ms_per_step": round(x["t_step"] * 1e3, 5), "winner": k == best} for k, x in runs.items()}
                table[state["trying"]] = {"error": f"did not finish within {args.exchange_deadline} s (hung or raised on some rank: see stderr)"}
                line = {"metric": "contrastive-pairs/sec (fwd+bwd)", "value": (b * world) ** 2 / r["t_step"], "unit": "pairs/s",
                        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["t_step"] * 1e3,
                        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                        "config": {"workload": f"CrossCLR_onlyIntraModality fwd+bwd, b={b} rows/GPU, global B={b * world}, D={d}, tau={TAU}, "
                                               f"negative_weight={NEG_W}, bf16 operands / fp32 accumulate (REDUCED LINE: an operand-exchange "
                                               f"form failed, the per-kernel section was not reached)",
                                   "global_batch": b * world, "rows_per_gpu": b, "dim": d, "parallelism": f"row-sharded x{world}",
                                   "operand_exchange": best},
                        "loss": r["loss"], "per_rank": r["per_rank"], "per_exchange": table}
                os.write(result_fd, (json.dumps(line) + "\n").encode())
            os._exit(0)

        for xm in ("allgather", "p2p", "p2p_each"):
            L.set_exchange_for_benchmark(xm)
            state["trying"] = xm
            dog = None
            if runs and args.exchange_deadline > 0:
                dog = threading.Timer(args.exchange_deadline, give_up)
                dog.daemon = True
                dog.start()
            try:
                if os.environ.get("CROSSCLR_BENCH_INJECT_FAILURE") == xm and rank == world - 1:     # (tests only)
                    raise RuntimeError(f"injected failure in the {xm} exchange")
                r = runs[xm] = timed_run()
            except Exception:
                if dog is None:
                    raise
                import traceback
                traceback.print_exc()
                threading.Event().wait()     # (the peers hang in their collectives: every rank leaves through its watchdog)
            if dog is not None:
                dog.cancel()
            exposed = [pr.get("exposed_comm_ms") for pr in (r["per_rank"] or []) if pr.get("exposed_comm_ms") is not None]
            per_exchange[xm] = {"ms_per_step": round(r["t_step"] * 1e3, 5),
                                "ms_per_step_event_median_rank0": round(r["ev_ms"][len(r["ev_ms"]) // 2], 5) if r["ev_ms"] else None,
                                "exposed_comm_ms_max_over_ranks": max(exposed) if exposed else None}
        best = min(per_exchange, key=lambda k: per_exchange[k]["ms_per_step"])
        L.set_exchange_for_benchmark(best)      # (the secondary / sustained measurements below run the winner)
        run = runs[best]
        for k in per_exchange:
            per_exchange[k]["winner"] = k == best
    else:
        run = timed_run()
    t_step, loss_val, ev_ms, per_rank = run["t_step"], run["loss"], run["ev_ms"], run["per_rank"]
    B = b * world

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    what = "fwd" if args.fwd_only else "fwd+bwd"
    if emu:   # tests only: the plumbing, not a measurement
        out = {"metric": f"contrastive-pairs/sec ({what})", "value": B * B / t_step, "unit": "pairs/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": t_step * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.mode == "bf16" else "f32", "data": "synthetic",
               "config": {"workload": "SELFTEST on the host emulation of the kernels (not a measurement)", "global_batch": B},
               "loss": loss_val, "per_rank": per_rank, "per_exchange": per_exchange}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # (the timed steps were the last launches so far: which kernel templates the library took for them)
    _lib = nat.library()
    step_kernels = [(_lib.crossclr_last_kernel(i) or b"").decode() for i in (0, 1)]
    # ---- per-kernel HIP-event timing of the single-GPU stages (roofline of the dominant kernel) ----
    sw = crossclr_amd.influential_sample_weights(xv, xt, 0.9, 0.0035) if args.influential else (None, None)
    st = _profile.stage_times(v.detach(), t.detach(), TAU, NEG_W, args.mode, iters=20, warmup=3,
                              negative_scale=sw[0], loss_weight=sw[1], in_step=50, forward_only=args.fwd_only)
    peak = PEAK_BF16_TFLOPS if args.mode == "bf16" else PEAK_F32_TFLOPS
    # algorithmic flops per launch: forward 6*b*b*D, backward 8*b*b*D for the local block
    # forward_save / backward_saved are what a training step launches when the plan has the save-for-backward pair
    # (forward / backward are the recomputing entry points, timed for comparison)
    alg = {"forward": 6.0 * b * b * d, "backward": 8.0 * b * b * d, "forward_save": 6.0 * b * b * d, "backward_saved": 8.0 * b * b * d,
           "backward_saved_lds": 8.0 * b * b * d, "backward_saved_xf1": 8.0 * b * b * d}
    kernels = {}
    # (normalize_plain / backward_saved_lds: the pair without the fragment-major operand copy, timed beside the one the step runs)
    # (backward_saved_xf1 / backward_saved_lds: the one-tile-per-barrier fragment-major kernel and the LDS-staged one, beside the step's)
    for k in ("normalize", "normalize_plain", "forward", "forward_save", "forward_finish", "backward", "backward_saved", "backward_saved_xf1",
              "backward_saved_lds", "backward_finish"):
        if k not in st:
            continue
        kernels[k] = {"ms": round(st[k], 4)}
        if k in alg:
            tf = alg[k] / (st[k] * 1e-3) / 1e12
            kernels[k].update(algorithmic_tflops=round(tf, 2), frac=round(tf / peak, 4))
            if k in ("forward", "forward_save") and world == 1:
                # the forward evaluates the upper triangle of the stacked 2b x 2b matrix only (row blocks of 256 / 128 rows on the diagonal):
                # 4 b^2 D (1 + RB / 2b) executed for the 6 b^2 D the reference's three GEMMs count -- `frac` is ALGORITHMIC and can pass 1
                ex = 4.0 * b * b * d * (1.0 + (256.0 if st["fast_path"] else 128.0) / (2.0 * b)) / (st[k] * 1e-3) / 1e12
                kernels[k].update(executed_tflops=round(ex, 2), frac_executed=round(ex / peak, 4))
    saved = bool(st.get("saved_path"))
    # what the library itself launched for the step's forward / gradient product (crossclr_last_kernel, ABI 7; asked right after the step's own
    # passes, before the recomputing / alternative entry points of the per-kernel table run)
    launched = {"forward": step_kernels[0], "gradient_product": step_kernels[1]}
    if args.fwd_only:
        # (whole 128-row batches without sample weights take round 5's forward: csrc/crossclr_kernels_symp.h)
        dom, dom_kernel = "forward", (launched["forward"] or ("fast_fwd_pipe_kernel" if st["fast_path"] else "fwd_sums_kernel"))
    else:
        dom = "backward_saved" if saved else "backward"
        dom_kernel = (("fast_bwd_dsl_kernel" if st["fast_path"] else "bwd_saved32_kernel") if saved
                      else ("fast_bwd" if st["fast_path"] else "bwd_kernel"))
    # the dominant kernel's duration IN ITS PLACE in the step (median over 50 passes of the step's kernel sequence, events between the
    # launches); `kernels[dom].ms` keeps the back-to-back figure (the kernel alone at the package power limit: a few per cent slower)
    in_step_key = "in_step_forward" if args.fwd_only else "in_step_backward"
    dom_ms = st.get(in_step_key, st[dom])
    dom_tf = alg[dom] / (dom_ms * 1e-3) / 1e12
    for k_in, k_iso in (("in_step_normalize", "normalize"), ("in_step_forward", "forward" if (args.fwd_only or not saved) else "forward_save"),
                        ("in_step_forward_finish", "forward_finish"), ("in_step_backward", "backward_saved" if saved else "backward"),
                        ("in_step_backward_finish", "backward_finish")):
        if k_in in st and k_iso in kernels:
            kernels[k_iso]["in_step_ms"] = round(st[k_in], 4)
    if "in_step_total" in st:
        kernels["in_step_sequence_total"] = {"ms": round(st["in_step_total"], 4)}
    # the saved backward of the local block has three kernels per width: the pair kernel on the fragment-major operand (fast_bwd_xfp_kernel:
    # what the step runs when stage_times reports xfp_path), the one-tile-per-barrier kernel on the same operand (fast_bwd_dsl_kernel<...,
    # true>) and the one that stages the column tiles through LDS (..., false>)
    dom_suffix = None
    entry_suffix = ""
    if dom_kernel == "fast_bwd_dsl_kernel" and st.get("xfp_path"):
        dom_kernel = "fast_bwd_xfp_kernel"
        dom_kernel_label = "fast_bwd_xfp_kernel: column tiles as MFMA fragments from the fragment-major operand, two tiles per barrier interval"
        entry_suffix = "_xfp"
    elif dom_kernel == "fast_bwd_dsl_kernel":
        dom_suffix = ", true>" if st.get("xf_path") else ", false>"
        dom_kernel_label = dom_kernel + ("<..., XF>: column tiles as MFMA fragments from the fragment-major operand" if st.get("xf_path") else "")
        entry_suffix = "_xf" if st.get("xf_path") else ""
    else:
        dom_kernel_label = dom_kernel
    traffic = measured_traffic(b, d, args.mode, dom_kernel, dom_suffix) if world == 1 and not args.influential else None
    step_tf = (6.0 if args.fwd_only else 14.0) * b * B * d / t_step / 1e12  # per-GPU algorithmic flops over the whole step
    out = {
        "metric": f"contrastive-pairs/sec ({what})", "value": B * B / t_step, "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": max(0, args.prewarm),
        "settle_steps_used": settle_used, "settle_chunk_ms_per_step": [round(x, 5) for x in settle_chunks[-6:]],
        "ms_per_step": t_step * 1e3,
        "ms_per_step_event_median": ev_ms[len(ev_ms) // 2] if ev_ms else None,
        "ms_per_step_event_min_max": [ev_ms[0], ev_ms[-1]] if ev_ms else None,
        "event_pass": "a second pass of the same K steps right after the timed window, HIP events around every step",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "methodology": "since round 5: the K timed steps start after an adaptive settle loop (`settle_steps_used`, three 20-step chunks within 1 %, "
                       "cap 400) and carry no per-step events (those are a second pass: `event_pass`); rounds 1-4 timed the K steps right after the "
                       "warm-up with events inside the window -- `value` is not comparable across that change (BENCH_r04 -> r05: 0.4649 -> 0.4183 "
                       "ms/step, of which ~0.02 is methodology, see DESIGN.md 3.3)",
        "dtype": "bf16" if args.mode == "bf16" else "f32", "data": "synthetic",
        "samples_per_s": B / t_step,
        "config": {"workload": (f"PLUMBING CHECK, NOT A MEASUREMENT ({world} ranks share ONE GPU, collectives over gloo): " if args.share_gpu else "") +
                               ("CrossCLR (influential-sample pruning + weighting) " if args.influential else "CrossCLR_onlyIntraModality ") +
                               f"{what}, b={b} rows/GPU, global B={B}, D={d}, "
                               f"tau={TAU}, negative_weight={NEG_W}, {args.mode} operands / fp32 accumulate, "
                               f"randn features seed 1234+rank, {max(0, args.prewarm)} untimed settle steps before the warm-up",
                   "global_batch": B, "rows_per_gpu": b, "dim": d,
                   "parallelism": f"row-sharded x{world}" + (" + RCCL all-gather of packed operands" if world > 1 else ""),
                   "fast_path": bool(st["fast_path"]), "save_for_backward": saved and not args.fwd_only},
        "loss": loss_val,
        "roofline": {"bound": "mfma", "kernel": f"{dom_kernel_label} (crossclr_{dom}{entry_suffix if dom == 'backward_saved' else ''}; dominant kernel)",
                     "achieved": round(dom_tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(dom_tf / peak, 4),
                     "source": "HIP events in this process (torch's current stream = the launch stream): median over 50 passes of the step's own "
                               "kernel sequence (normalize, forward + save, finish [through the stage entry point: with the separate loss-reduce launch "
                               "that crossclr_step_forward folds into the finish kernel], saved backward, finish) right after the timed region, events "
                               "between the launches -- the kernel in its place in the step, which is what the rocprofv3 trace of the same command "
                               "(profiles/) shows; `back_to_back_launch_ms` = the median of 20 launches of the kernel alone (package power limit: slower)",
                     "back_to_back_launch_ms": round(st[dom], 4),
                     "traffic": traffic["bytes"] if traffic else None,
                     "traffic_source": (traffic["source"] + " (rocprofv3 --pmc passes of tools/kbench.py, read from the committed "
                                        "summary -- bench.py cannot profile itself)") if traffic else None,
                     "traffic_csrc_sha": traffic["csrc_sha"] if traffic else None, "csrc_sha": csrc_sha(),
                     "traffic_stale": (traffic["csrc_sha"] != csrc_sha()) if traffic else None,
                     "hbm_gbps_at_that_traffic": round(traffic["bytes"] / (dom_ms * 1e-3) / 1e9, 1) if traffic else None,
                     "algorithmic_hbm_bytes_per_launch": 2.0 * b * d * 2 + 2.0 * b * d * 4 + 6.0 * b * 4,
                     "algorithmic_flops_per_launch": alg[dom], "avg_launch_ms": round(dom_ms, 4),      # (the in-step median)
                     "whole_step_algorithmic_tflops_per_gpu": round(step_tf, 2),
                     "whole_step_frac": round(step_tf / peak, 4),
                     },
        "kernels": kernels,
        "launched": launched,
    }
    if args.fwd_only and world == 1:
        ex = 4.0 * b * b * d * (1.0 + (256.0 if st["fast_path"] else 128.0) / (2.0 * b)) / (dom_ms * 1e-3) / 1e12
        out["roofline"].update(executed_tflops=round(ex, 2), frac_executed=round(ex / peak, 4),
                               note="symmetric forward: the upper triangle of the stacked matrix is evaluated, 4.06 b^2 D executed for the 6 b^2 D "
                                    "algorithmic flops -- `frac` (algorithmic, the contract's definition) can pass 1, `frac_executed` cannot")
    if args.mode == "bf16":
        # what a loop of nothing but MFMAs sustains on this device, measured NOW (the package sits at its power limit with toggling
        # operands: DESIGN.md 3.1) -- not a constant carried over from an earlier round
        sm = sustained_mfma(dev)
        out["roofline"]["sustained_mfma"] = sm
        out["roofline"]["frac_of_sustained_random_operands"] = round(dom_tf / sm["random_operands"]["tflops"], 4)
        if world == 1 and not args.fwd_only:
            lg = library_gemm(dev, 2 * b, d)
            out["roofline"]["library_gemm"] = lg
            out["roofline"]["frac_of_library_gemm"] = round(dom_tf / lg["tflops"], 4)
    if world == 1 and args.sustained_steps > 0:
        out["sustained"] = sustained_series(step, dev, args.sustained_steps)
    if per_rank is not None:
        # (the waits are measured on a handful of extra steps with events around every collective wait: `exposed_comm_ms` is
        #  communication the compute stream had to sit out, `compute_ms` the rest of that step)
        out["per_rank"] = per_rank
        out["config"]["operand_exchange"] = per_rank[0].get("exchange")
        if per_exchange is not None:
            # (three timed regions were run, one per way the operands can travel; `value` is the fastest one's)
            out["per_exchange"] = per_exchange
    if args.influential:
        out["config"]["pruned_fraction"] = [round(1.0 - float(k.mean()), 4) for k in sw[0]]
    if world == 1 and b == B_PER_GPU and d == DIM and not args.influential:
        out["loss_delta_vs_reference"] = abs(loss_val - GOLDEN_LOSS_B8192_SEED1234)
    if world == 1 and b == 4096 and d == DIM and not args.influential:
        out["loss_delta_vs_reference"] = abs(loss_val - GOLDEN_LOSS_B4096_SEED1234)
    if world == 1 and b == B_PER_GPU and d == DIM and args.mode == "bf16" and not args.influential and not args.fwd_only and not args.no_secondary:
        out["secondary"] = secondary_lines(dev)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(b, d, args.fwd_only)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
