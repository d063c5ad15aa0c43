import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, crossclr_amd
from crossclr_amd import loss as L
from oracle import crossclr_oracle as orc
B, D = 8192, 512
v, t = orc.make_inputs("randn", B, D, 99)
vd, td = v.cuda(), t.cuda()
ref = None
for it in range(40):
    save = it % 2 == 1
    loss, ws = L._forward_impl(vd, td, 0.03, 0.8, "bf16", None, save_for_backward=save)
    torch.cuda.synchronize()
    lz = ws.logz.clone()
    if ref is None:
        ref = lz
    bad = (~torch.isfinite(lz)) | ((lz - ref).abs() > 1e-3)
    n = int(bad.sum())
    if n:
        idx = bad.nonzero().flatten()
        print(f"iter {it} save={save}: loss {loss.item()} bad rows {n}: first {idx[:8].tolist()} last {idx[-4:].tolist()} blocks256 {sorted(set((idx // 256).tolist()))[:20]}")
    else:
        print(f"iter {it} save={save}: loss {loss.item()} ok")
