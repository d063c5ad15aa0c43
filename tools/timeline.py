#!/usr/bin/env python3
"""Block timeline of the two pipelined kernels (BASELINE config 3 by default): when each of the 256 blocks starts, has its first tile,
leaves its main loop and exits, and the shader clock the launch ran at.  Needs a library built with -DCROSSCLR_TIMING:
    python tools/build_variant.py tm -DCROSSCLR_TIMING -DCROSSCLR_DSL_MINIMAL
    CROSSCLR_HIP_LIBRARY=variants/libtm.so python tools/timeline.py [B] [D]
The marks are s_memrealtime (100 MHz, one counter for the whole device) -> 10 ns resolution."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from crossclr_amd import _native as nat, loss as L
from oracle import crossclr_oracle as orc

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = nat.library()
lib.crossclr_debug_timing.restype = ctypes.c_int
lib.crossclr_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_int]
v, t = orc.make_inputs("randn", B, D, 1234)
v, t = v.cuda(), t.cuda()
_, ws = L._forward_impl(v, t, 0.03, 0.8, "bf16", None, None, None, save_for_backward=True)
plan, pp, p = ws.plan, ctypes.byref(ws.plan), L._ptr
stream = L._stream_for(v)
part = torch.empty(plan.fwd_ws_floats, dtype=torch.float32, device="cuda")
gbuf = torch.empty(plan.gbuf_bytes // 4, dtype=torch.float32, device="cuda")
stages = {
    "forward_save": lambda: lib.crossclr_forward_save(pp, p(ws.xhat), 0.03, 0.8, None, p(part), 0, p(ws.stash), stream),
    "backward_saved": lambda: lib.crossclr_backward_saved(pp, p(ws.xhat), p(ws.stash), 0.03, 0.8, p(ws.rz), p(ws.wrz), None, p(gbuf), 0, stream),
}
if ws.xf is not None:       # the step's own backward: the pair kernel on the fragment-major operand
    stages["backward_saved_xfp"] = lambda: lib.crossclr_backward_saved_xfp(pp, p(ws.xf), p(ws.stash), 0.03, 0.8, p(ws.rz), p(ws.wrz), None, p(gbuf), 0, stream)
TICK = 10.0   # ns per s_memrealtime tick


def pct(x, q):
    return float(np.percentile(x, q))


for name, fn in stages.items():
    for _ in range(400):           # settle the clock / power state: the marks of the LAST launch are read
        nat.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); nat.check(fn()); e1.record()
    torch.cuda.synchronize()
    raw = np.zeros((256, 8), dtype=np.uint64)
    nat.check(lib.crossclr_debug_timing(raw.ctypes.data, 256))
    raw = raw[raw[:, 0] != 0]       # (thread blocks the launch did not have leave their rows untouched)
    nb = raw.shape[0]
    m = raw[:, :4].astype(np.int64)
    t0 = m[:, 0].min()
    start, first, loop_end, end = [(m[:, k] - t0) * TICK / 1000.0 for k in range(4)]      # us
    clk = (raw[:, 5].astype(np.int64) - raw[:, 4].astype(np.int64)) / np.maximum((m[:, 3] - m[:, 0]) * TICK, 1) # cycles per ns = GHz
    xcc = (raw[:, 7] & 0xF).astype(np.int64)
    print(f"== {name}: HIP events {e0.elapsed_time(e1) * 1000:.1f} us; first block start -> last block exit {end.max():.1f} us; "
          f"shader clock {np.median(clk):.3f} GHz (p5 {pct(clk, 5):.3f}, p95 {pct(clk, 95):.3f})")
    for label, x in (("block start", start), ("first tile ready - start", first - start), ("main loop", loop_end - first),
                     ("tail (stores) ", end - loop_end), ("block exit", end), ("block lifetime", end - start)):
        print(f"   {label:26s}: min {x.min():7.2f}  p25 {pct(x, 25):7.2f}  median {pct(x, 50):7.2f}  p75 {pct(x, 75):7.2f}  max {x.max():7.2f} us")
    busy = (end - start).sum() / (nb * end.max())
    print(f"   average block residency over the launch: {busy:.3f}; XCC ids seen: {sorted(set(xcc.tolist()))}")
    for x in sorted(set(xcc.tolist())):
        sel = xcc == x
        print(f"     XCC {x}: {sel.sum():3d} blocks, start {start[sel].min():6.2f}..{start[sel].max():6.2f}, exit {end[sel].min():7.2f}..{end[sel].max():7.2f} us, "
              f"main loop median {pct((loop_end - first)[sel], 50):7.2f}, shader clock {np.median(clk[sel]):.3f} GHz")
    if name.startswith("backward_saved") and nb == 256:      # block = row block x (128 rows) + 128 * column slice y
        ml = (loop_end - first).reshape(2, 128)
        for y in range(2):
            print(f"     slice {y}: main loop by row block, means of 16: " + " ".join(f"{ml[y, k:k + 16].mean():6.1f}" for k in range(0, 128, 16)))
    order = np.argsort(end)
    print("   last blocks to exit (block: start, main loop, exit):", ", ".join(f"{b}: {start[b]:.1f}/{(loop_end - first)[b]:.1f}/{end[b]:.1f}" for b in order[-6:]))
    print("   first blocks to exit                              :", ", ".join(f"{b}: {start[b]:.1f}/{(loop_end - first)[b]:.1f}/{end[b]:.1f}" for b in order[:6]))
