import sys, torch, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crossclr_amd
from oracle import crossclr_oracle as orc
xv, xt = orc.make_inputs("cluster", 8192, 256, 4321)
xv, xt = xv.cuda(), xt.cuda()
for _ in range(5): crossclr_amd.influential_sample_weights(xv, xt, 0.9, 0.0035)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): crossclr_amd.influential_sample_weights(xv, xt, 0.9, 0.0035)
e1.record(); torch.cuda.synchronize()
print("influential_sample_weights: %.1f us (gpu), " % (e0.elapsed_time(e1) * 20))
t0 = time.perf_counter()
for _ in range(50): crossclr_amd.influential_sample_weights(xv, xt, 0.9, 0.0035)
torch.cuda.synchronize()
print("wall per call %.1f us" % ((time.perf_counter() - t0) * 2e4))
