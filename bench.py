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
between barrier + synchronize fences, max over ranks.  The settle steps exist because an MI355X that has just been
handed to the process runs its first ~20-50 steps ~9 % slower (clock ramp); they are reported as `prewarm_steps`.
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


def measured_traffic(b, d, mode, kernel_substr):
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
            if kernel_substr in name:
                best = {"bytes": e["hbm_bytes_corrected"], "source": os.path.relpath(path, ROOT)}
    return best


def cpu_baseline(b, d):
    """The oracle's op-for-op restatement of the reference (bit-identical to it, see
    tests/golden/make_golden.py) timed on this box's host cores: bounded sample."""
    from oracle import crossclr_oracle as orc
    try:
        import psutil
        avail_gb = psutil.virtual_memory().available / 2**30
    except Exception:
        avail_gb = 0.0
    bb = b if avail_gb >= 24 else min(b, 4096)
    v, t = orc.make_inputs("randn", bb, d, 1234)
    orc.eager_loss_and_grads(v[:512], t[:512], TAU, NEG_W)  # warm the allocator / thread pool
    times = []
    budget_t0 = time.perf_counter()
    for i in range(3):
        t0 = time.perf_counter()
        loss, _, _ = orc.eager_loss_and_grads(v, t, TAU, NEG_W)
        times.append(time.perf_counter() - t0)
        if time.perf_counter() - budget_t0 > 25 and i >= 1:
            break
    best = sorted(times[1:] or times)[len(times[1:] or times) // 2]
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "")
    except OSError:
        pass
    return {"value": bb * bb / best, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model,
            "samples_per_s": bb / best, "seconds_per_step": best, "host_cpus": os.cpu_count(),
            "loss": float(loss),
            "sample": f"oracle.eager_loss_and_grads (op-for-op reference restatement) fwd+bwd, fp32 inputs, "
                      f"B={bb} D={d}, {len(times)} steps (first = warm-up), median of the rest"}


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--influential", action="store_true",
                    help="BASELINE config 5: influential-sample pruning + loss weighting from synthetic input-space "
                         "features (crossclr_amd.CrossCLR); not the default workload")
    args = ap.parse_args()

    # RCCL prints a version banner on the C-level stdout of every rank.  Keep this process's stdout to
    # exactly ONE JSON line: everything else that lands on fd 1 is sent to stderr; the result goes to
    # the saved descriptor.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    import crossclr_amd
    from crossclr_amd import _native as nat
    from crossclr_amd import _profile
    from oracle import crossclr_oracle as orc

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    assert nat.backend() == "hip-gfx950", "bench needs the HIP library"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
        group = dist.group.WORLD

    b, d = args.rows, args.dim
    v, t = orc.make_inputs("randn", b, d, 1234 + rank)
    v = v.to(dev).requires_grad_(True)
    t = t.to(dev).requires_grad_(True)
    if args.influential:
        # input-space features: 16 clusters + noise (so that connectivities differ and the threshold prunes a part)
        xv, xt = orc.make_inputs("cluster", b, 256, 4321 + rank)
        xv, xt = xv.to(dev), xt.to(dev)
        crit = crossclr_amd.CrossCLR(TAU, 0.0035, NEG_W, 0.9, compute_mode=args.mode, process_group=group).to(dev)
    else:
        crit = crossclr_amd.CrossCLR_onlyIntraModality(TAU, NEG_W, compute_mode=args.mode, process_group=group).to(dev)

    def step():
        v.grad = None
        t.grad = None
        loss = crit(v, t, xv, xt) if args.influential else crit(v, t)   # the O(B D) weight recipe is inside the step
        loss.backward()
        return loss

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(max(0, args.prewarm)):
        step()
    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = tmax.item()
    t_step = elapsed / args.steps
    B = b * world
    loss_val = float(loss.item())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- per-kernel HIP-event timing of the single-GPU stages (roofline of the dominant kernel) ----
    sw = crossclr_amd.influential_sample_weights(xv, xt, 0.9, 0.0035) if args.influential else (None, None)
    st = _profile.stage_times(v.detach(), t.detach(), TAU, NEG_W, args.mode, iters=10, warmup=2,
                              negative_scale=sw[0], loss_weight=sw[1])
    peak = PEAK_BF16_TFLOPS if args.mode == "bf16" else PEAK_F32_TFLOPS
    # algorithmic flops per launch: forward 6*b*b*D, backward 8*b*b*D for the local block
    alg = {"forward": 6.0 * b * b * d, "backward": 8.0 * b * b * d}
    kernels = {}
    for k in ("normalize", "forward", "forward_finish", "backward", "backward_finish"):
        kernels[k] = {"ms": round(st[k], 4)}
        if k in alg:
            tf = alg[k] / (st[k] * 1e-3) / 1e12
            kernels[k].update(algorithmic_tflops=round(tf, 2), frac=round(tf / peak, 4))
    dom = "backward"
    dom_tf = alg[dom] / (st[dom] * 1e-3) / 1e12
    traffic = (measured_traffic(b, d, args.mode, "fast_bwd_kernel" if st["fast_path"] else "bwd_kernel")
               if world == 1 and not args.influential else None)
    step_tf = 14.0 * b * B * d / t_step / 1e12  # per-GPU algorithmic fwd+bwd flops over the whole step
    out = {
        "metric": "contrastive-pairs/sec (fwd+bwd)", "value": B * B / t_step, "unit": "pairs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": max(0, args.prewarm),
        "ms_per_step": t_step * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if args.mode == "bf16" else "f32", "data": "synthetic",
        "samples_per_s": B / t_step,
        "config": {"workload": ("CrossCLR (influential-sample pruning + weighting) " if args.influential else "CrossCLR_onlyIntraModality ") +
                               f"fwd+bwd, b={b} rows/GPU, global B={B}, D={d}, "
                               f"tau={TAU}, negative_weight={NEG_W}, {args.mode} operands / fp32 accumulate, "
                               "randn features seed 1234+rank",
                   "global_batch": B, "rows_per_gpu": b, "dim": d,
                   "parallelism": f"row-sharded x{world}" + (" + RCCL all-gather of packed operands" if world > 1 else ""),
                   "fast_path": bool(st["fast_path"])},
        "loss": loss_val,
        "roofline": {"bound": "mfma", "kernel": "crossclr_backward (dominant kernel)",
                     "achieved": round(dom_tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(dom_tf / peak, 4),
                     "traffic": traffic["bytes"] if traffic else None,
                     "traffic_source": traffic["source"] if traffic else None,
                     "hbm_gbps_at_that_traffic": round(traffic["bytes"] / (st[dom] * 1e-3) / 1e9, 1) if traffic else None,
                     "algorithmic_hbm_bytes_per_launch": 2.0 * b * d * 2 + 2.0 * b * d * 4 + 6.0 * b * 4,
                     "algorithmic_flops_per_launch": alg[dom], "avg_launch_ms": round(st[dom], 4),
                     "whole_step_algorithmic_tflops_per_gpu": round(step_tf, 2),
                     "whole_step_frac": round(step_tf / peak, 4)},
        "kernels": kernels,
    }
    if args.influential:
        out["config"]["pruned_fraction"] = [round(1.0 - float(k.mean()), 4) for k in sw[0]]
    if world == 1 and b == B_PER_GPU and d == DIM and not args.influential:
        out["loss_delta_vs_reference"] = abs(loss_val - GOLDEN_LOSS_B8192_SEED1234)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(b, d)
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
