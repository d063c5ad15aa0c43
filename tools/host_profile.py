#!/usr/bin/env python3
"""Host-side cost of a training step (enqueue only; the device runs behind): cProfile of 2000 steps of module forward + backward at a size
where the device is faster than the host would need it to be.  usage: host_profile.py [B] [D] [top]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
g = torch.Generator().manual_seed(1)
v = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
t = torch.randn(B, D, generator=g).cuda().requires_grad_(True)
crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
def step():
    v.grad = t.grad = None
    crit(v, t).backward()
for _ in range(200): step()
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for _ in range(N): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"B={B} D={D}: host enqueue {1e6 * (t1 - t0) / N:.1f} us per step (device behind: {1e6 * (time.perf_counter() - t0) / N:.1f} us per step wall)")
fwd = bwd = 0.0
for _ in range(N):
    v.grad = t.grad = None
    a = time.perf_counter(); l = crit(v, t); b = time.perf_counter(); l.backward(); c = time.perf_counter()
    fwd += b - a; bwd += c - b
torch.cuda.synchronize()
print(f"forward() {1e6 * fwd / N:.1f} us, backward() {1e6 * bwd / N:.1f} us")
# the two C calls alone (ctypes marshalling + the library's launches)
lib = crossclr_amd._native.library()
acc = {"crossclr_step_forward": 0.0, "crossclr_step_backward": 0.0}
orig = {k: getattr(lib, k) for k in acc}
def wrap(name):
    f = orig[name]
    def g(*a):
        t0 = time.perf_counter(); r = f(*a); acc[name] += time.perf_counter() - t0; return r
    return g
for k in acc: setattr(lib, k, wrap(k))
for _ in range(N): step()
torch.cuda.synchronize()
for k in acc: setattr(lib, k, orig[k])
print("inside the C calls: " + ", ".join(f"{k} {1e6 * v / N:.1f} us" for k, v in acc.items()))
pr = cProfile.Profile(); pr.enable()
for _ in range(N): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(top)
