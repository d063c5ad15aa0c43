#!/usr/bin/env python3
"""Per-kernel timing at one shape (default: BASELINE config 3).  usage: kbench.py [B] [D] [mode]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from crossclr_amd import _profile
from oracle import crossclr_oracle as orc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mode = sys.argv[3] if len(sys.argv) > 3 else "bf16"
v, t = orc.make_inputs("randn", B, D, 1234)
v, t = v.cuda(), t.cuda()
st = _profile.stage_times(v, t, 0.03, 0.8, mode, iters=20, warmup=3, settle=int(os.environ.get("KBENCH_SETTLE", "30")))
peak = 2500.0 if mode == "bf16" else 157.3
f = 6.0 * B * B * D / (st["step_forward"] * 1e-3) / 1e12
b = 8.0 * B * B * D / (st["step_backward"] * 1e-3) / 1e12
tot = sum(st[k] for k in ("normalize", "step_forward", "forward_finish", "step_backward", "backward_finish"))
print(f"B={B} D={D} {mode} fast={int(st['fast_path'])} saved={int(st['saved_path'])} xf={int(st.get('xf_path', 0))} xfp={int(st.get('xfp_path', 0))}: " + " ".join(f"{k}={st[k]:.4f}ms" for k in st if k not in ("fast_path", "saved_path", "xf_path", "xfp_path", "step_forward", "step_backward")) +
      f" | sum={tot:.4f}ms fwd {f:.0f} TF alg ({f/peak:.1%}) bwd {b:.0f} TF alg ({b/peak:.1%}) step {14.0*B*B*D/(tot*1e-3)/1e12/peak:.1%}")
