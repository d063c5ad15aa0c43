#!/usr/bin/env python3
"""Soak of the training step: STEPS steps over three alternating batches on one workspace stream, no host synchronisation in between; every
loss and a checksum of both gradients must repeat the first evaluation of its batch BIT FOR BIT (hand-counted waits, ring reuse, the
ticket hand-off of the loss reduce, clock ramps: anything that depends on timing shows up as a flipped bit sooner or later).
usage: soak_step.py [B] [D] [STEPS] [weighted 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
weighted = len(sys.argv) > 4 and sys.argv[4] == "1"
g = torch.Generator().manual_seed(77)
batches = []
for k in range(3):
    v = torch.randn(B, D, generator=g)
    t = (0.2 + 0.3 * k) * v + torch.randn(B, D, generator=g)
    batches.append((v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)))
crit = (crossclr_amd.CrossCLR(0.03, negative_weight=0.8, compute_mode="bf16") if weighted
        else crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16")).cuda()
inputs = [(torch.randn(B, 64, generator=g).cuda(), torch.randn(B, 64, generator=g).cuda()) for _ in range(3)]   # input-space features (weights)


def step(k):
    v, t = batches[k]
    v.grad = t.grad = None
    loss = crit(v, t, *inputs[k]) if weighted else crit(v, t)
    loss.backward()
    # checksum: the gradients' bits as int32, summed with wrap-around (order-independent, exact)
    cs = v.grad.view(torch.int32).sum(dtype=torch.int64) * 3 + t.grad.view(torch.int32).sum(dtype=torch.int64)
    return loss.detach().clone(), cs


first = [step(k) for k in range(3)]
torch.cuda.synchronize()
t0 = time.perf_counter()
bad = 0
CH = 1000
done = 0
while done < STEPS:
    got = [(i % 3, *step(i % 3)) for i in range(done, min(STEPS, done + CH))]
    torch.cuda.synchronize()
    for k, l, cs in got:
        if l.item() != first[k][0].item() or cs.item() != first[k][1].item():
            bad += 1
            if bad <= 5:
                print(f"MISMATCH batch {k}: loss {l.item()!r} vs {first[k][0].item()!r}, checksum {cs.item()} vs {first[k][1].item()}", flush=True)
    done += len(got)
dt = time.perf_counter() - t0
print(f"soak B={B} D={D} weighted={int(weighted)}: {STEPS} steps over 3 alternating batches in {dt:.1f} s ({1e3 * dt / STEPS:.3f} ms/step incl. checksums), "
      f"losses {[round(f[0].item(), 6) for f in first]}: {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
