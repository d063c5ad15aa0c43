#!/bin/bash
# usage: tools/clock_probe.sh: samples rocm-smi clocks / power while the default step (B=8192, D=512, bf16) loops
python - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, crossclr_amd
from bench import make_inputs
v, t = make_inputs(8192, 512, 1234)
v, t = v.cuda().requires_grad_(True), t.cuda().requires_grad_(True)
crit = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()
t0 = time.time()
while time.time() - t0 < 25:
    for _ in range(200):
        v.grad = t.grad = None
        crit(v, t).backward()
    torch.cuda.synchronize()
PY
PID=$!
sleep 12
for i in 1 2 3 4; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" | tr '\n' ';'; echo
  sleep 2
done
wait $PID
echo idle:; rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | tr '\n' ';'; echo
