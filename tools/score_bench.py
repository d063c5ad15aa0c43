#!/usr/bin/env python3
"""Timing of the score-statistics entry points (retrieval ranks / max-margin forward and backward: the recomputing backward and the one from
the saved hinge mask).  usage: score_bench.py [B] [D]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crossclr_amd import _native as nat, loss as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib, p = nat.library(), L._ptr
g = torch.Generator().manual_seed(1)
v = torch.randn(B, D, generator=g).cuda(); t = torch.randn(B, D, generator=g).cuda()
for mode, name in ((nat.MODE_FP32, "fp32"), (nat.MODE_BF16, "bf16")):
    plan = nat.make_plan(B, D, 1, 0, mode); pp = ctypes.byref(plan)
    st = L._stream_for(v); f32 = dict(dtype=torch.float32, device="cuda")
    x = torch.empty(plan.operand_bytes, dtype=torch.uint8, device="cuda"); inv = torch.empty(2 * plan.bpad, **f32); dg = torch.empty(plan.bpad, **f32)
    nat.check(lib.crossclr_normalize(pp, p(v), p(t), D, D, nat.IN_F32, p(x), p(inv), p(dg), st))
    diag = torch.empty(2 * plan.bpad, **f32); part = torch.empty(plan.fwd_ws_floats, **f32)
    hinge, act = torch.empty(2 * plan.bpad, **f32), torch.empty(2 * plan.bpad, **f32)
    ls = torch.empty(plan.loss_ws_doubles, dtype=torch.float64, device="cuda")
    def timeit(fn):
        for _ in range(3): nat.check(fn())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): nat.check(fn())
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    td = timeit(lambda: lib.crossclr_score_diag(pp, p(x), p(diag), st))
    tr = timeit(lambda: lib.crossclr_score_rows(pp, p(x), p(diag), 0.0, p(part), p(hinge), p(act), p(ls), st))
    mask = torch.empty(lib.crossclr_maxmargin_mask_bytes(pp), dtype=torch.uint8, device="cuda")
    gbuf = torch.empty(plan.gbuf_bytes // 4, **f32)
    ts = timeit(lambda: lib.crossclr_score_rows_save(pp, p(x), p(diag), 0.1, p(part), p(hinge), p(act), p(ls), p(mask), st))
    tb = timeit(lambda: lib.crossclr_maxmargin_backward(pp, p(x), p(diag), 0.1, p(gbuf), st))
    g0 = gbuf.clone()
    tbs = timeit(lambda: lib.crossclr_maxmargin_backward_saved(pp, p(x), p(mask), p(gbuf), st))
    print(f"B={B} D={D} {name}: fwd_slots={plan.fwd_slots} score_diag {td:.3f} ms, score_rows {tr:.3f} ms (+ hinge mask: {ts:.3f}); max-margin backward "
          f"recomputing {tb:.3f} ms, from the saved mask {tbs:.3f} ms ({tb / tbs:.2f}x; forward + backward {tr + tb:.3f} -> {ts + tbs:.3f} ms), same bits: {torch.equal(g0, gbuf)}")
