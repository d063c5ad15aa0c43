import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, crossclr_amd
B, D, K = 8192, 512, 20
g = torch.Generator().manual_seed(1234)
v = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
t = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
def step():
    v.grad = t.grad = None
    l = crit(v, t); l.backward(); return l
for _ in range(400): step()
for trial in range(5):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    es, ee = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    es.record()
    for i in range(K):
        ev[i][0].record(); step(); ev[i][1].record()
    ee.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    span = es.elapsed_time(ee)
    evs = [a.elapsed_time(z) for a, z in ev]
    gaps = [ev[i][1].elapsed_time(ev[i + 1][0]) for i in range(K - 1)]
    print(f"wall {1e3*(t2-t0):.3f} ms  host-enqueue {1e3*(t1-t0):.3f}  gpu span {span:.3f}  sum(ev) {sum(evs):.3f}  wall-span {1e3*(t2-t0)-span:.3f}  first-kernel delay {es.elapsed_time(ev[0][0]):.3f}")
    print("   ev[0..3]", " ".join(f"{x:.4f}" for x in evs[:4]), " ev[-2:]", " ".join(f"{x:.4f}" for x in evs[-2:]), " median", f"{sorted(evs)[K//2]:.4f}", " gaps max", f"{max(gaps):.4f}", "sum", f"{sum(gaps):.4f}")
# sync wake-up latency: a tiny kernel, then synchronize
x = torch.zeros(1, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); x.add_(1); torch.cuda.synchronize(); print(f"tiny kernel + sync: {1e6*(time.perf_counter()-t0):.1f} us")
