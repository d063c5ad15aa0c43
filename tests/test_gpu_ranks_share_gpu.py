"""Several RANKS (separate processes) of a row-sharded run on ONE MI355X: the product host path (`loss.py`: operand exchange, pair
scheme, partner gradients, the per-rank workspaces) with the REAL HIP kernels, one process per rank as in production, the
collectives over `gloo` on device tensors -- RCCL refuses two ranks on one device (`profiles/r04l_probe_nccl.txt`), gloo accepts
them (`profiles/r04l_probe_gloo.txt`).  What the other suites leave open and this one closes: `tests/test_distributed_cpu.py`
runs the same host code in separate processes but on the host emulation of the kernels; `tests/test_gpu_parity.py` runs the real
kernels but plays all ranks from one process through the C-ABI.  Expected values: the float64 streaming oracle's global loss
and the gradient of the GLOBAL loss w.r.t. the rank's own rows (reference: /root/reference/trainer/loss.py:76-114 on the
concatenated batch).  Bars: BASELINE.json north_star (|loss - ref| <= 1e-3, gradients within 1e-2 of max|grad|) for bf16 at
tau = 0.03; exact fp32 much tighter."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, b, D, mode, tau, q):
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        mode, *knobs = mode.split("+")
        for k in knobs:
            if k in ("allgather", "p2p", "p2p_each"):
                os.environ["CROSSCLR_EXCHANGE"] = k
            if k == "recompute":
                os.environ["CROSSCLR_PARTNER_GRADS"] = "0"
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 8) // world)))      # (the float64 oracle below: no oversubscribed host)
        import crossclr_amd
        from crossclr_amd import _native as nat
        from crossclr_amd import loss as L
        from oracle import crossclr_oracle as orc
        assert nat.backend() == "hip-gfx950", "GPU tests must run the HIP library"
        B = b * world
        v, t = orc.make_inputs("randn", B, D, 4242)
        vl = v[rank * b:(rank + 1) * b].cuda().requires_grad_(True)
        tl = t[rank * b:(rank + 1) * b].cuda().requires_grad_(True)
        crit = crossclr_amd.CrossCLR_onlyIntraModality(tau, 0.8, compute_mode=mode, process_group=dist.group.WORLD).cuda()
        losses = []
        for _ in range(3):      # three steps: reused workspaces / exchange buffers must give the same answer every time
            vl.grad = None
            tl.grad = None
            loss = crit(vl, tl)
            loss.backward()
            torch.cuda.synchronize()
            losses.append(float(loss))
        assert max(losses) - min(losses) == 0.0, losses
        with torch.no_grad():
            again = float(crit(vl, tl))
        assert abs(again - losses[0]) <= 2e-6 * max(1.0, abs(losses[0])), (again, losses[0])
        ref = orc.sharded_loss_and_grads(v, t, world, rank, tau, 0.8)
        scale = ref["grad_v"].abs().max().item()
        q.put((rank, losses[0], float(ref["loss"]),
               (vl.grad.double().cpu() - ref["grad_v"]).abs().max().item() / scale,
               (tl.grad.double().cpu() - ref["grad_t"]).abs().max().item() / scale, L._last_exchange_mode))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # surface the failure in the parent
        import traceback
        q.put((rank, "error", traceback.format_exc(), 0, 0, None))


@pytest.mark.parametrize("world,b,D,mode,tau,ltol,gtol", [
    (2, 2048, 512, "bf16", 0.03, 1e-3, 1e-2),                 # antipodal block evaluated by both ranks, all-gather
    (3, 512, 512, "bf16", 0.03, 1e-3, 1e-2),                  # pair scheme + partner gradients, default exchange (p2p)
    (4, 512, 512, "bf16+p2p_each", 0.03, 1e-3, 1e-2),         # one launch per partner as its slice lands + the antipode
    (4, 512, 256, "bf16+allgather", 0.03, 1e-3, 1e-2),
    (3, 512, 512, "bf16+recompute", 0.03, 1e-3, 1e-2),        # the partner recomputes the pair block
    (5, 256, 128, "bf16", 0.03, 1e-3, 1e-2),                  # two pairs with wrap-around
    (2, 512, 512, "fp32", 0.03, 2e-6, 2e-5),                  # exact fp32: remote block from saved exponentials
    (3, 256, 512, "fp32", 0.005, 1e-5, 1e-4),                 # two-pass regime (row maxima exchanged between the passes)
    (3, 512, 512, "bf16", 0.005, 6e-3, 2e-2),                 # bf16 in the two-pass regime (logit quantisation ~ 1/tau: relaxed as in test_gpu_parity)
    (2, 512, 1536, "bf16", 0.03, 1e-3, 1e-2),                 # wide plan (D > 1024): generic forward that saves, D-slice backward
])
def test_ranks_in_separate_processes_on_one_gpu(world, b, D, mode, tau, ltol, gtol):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, b, D, mode, tau, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    losses = []
    for rank, loss, ref, ev, et, xm in sorted(results, key=lambda r: r[0]):
        assert loss != "error", ref
        assert abs(loss - ref) <= ltol * max(1.0, abs(ref)), (rank, loss, ref)
        assert ev <= gtol and et <= gtol, (rank, ev, et)
        losses.append(loss)
    assert max(losses) - min(losses) <= 1e-12, "every rank must see the same global loss"


def test_bench_multi_rank_path_with_the_real_kernels():
    """`bench.py --gpus 3 --share-gpu`: the launcher path, the barrier + synchronize fences, the max over ranks, the three-way
    operand-exchange table and the per-rank wait diagnostics, on the real kernels (timings meaningless: three ranks share the GPU)."""
    env = dict(os.environ)
    env.pop("CROSSCLR_EXCHANGE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--share-gpu", "--steps", "3", "--warmup", "1",
                        "--prewarm", "2", "--rows", "2048", "--no-cpu-baseline", "--no-secondary", "--sustained-steps", "0"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 3 and out["steps"] == 3 and out["config"]["global_batch"] == 3 * 2048
    assert "NOT A MEASUREMENT" in out["config"]["workload"]
    assert set(out["per_exchange"]) == {"allgather", "p2p", "p2p_each"}
    assert sum(1 for v in out["per_exchange"].values() if v["winner"]) == 1
    assert len(out["per_rank"]) == 3 and all(pr.get("exposed_comm_ms") is not None for pr in out["per_rank"])
    # the loss of the global batch: one process, one GPU, the same rows (seed 1234 + rank per rank, as bench.py draws them)
    sys.path.insert(0, ROOT)
    import bench
    import crossclr_amd
    vs, ts = zip(*[bench.make_inputs(2048, 512, 1234 + k) for k in range(3)])
    crit = crossclr_amd.CrossCLR_onlyIntraModality(bench.TAU, bench.NEG_W, compute_mode="bf16").cuda()
    with torch.no_grad():
        whole = float(crit(torch.cat(vs).cuda(), torch.cat(ts).cuda()))
    assert abs(out["loss"] - whole) <= 1e-4 * max(1.0, abs(whole)), (out["loss"], whole)
