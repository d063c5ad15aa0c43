#!/usr/bin/env python3
"""Does SQ_INSTS_VALU count the MFMA instructions themselves?  Launches the library's pure-MFMA loop (crossclr_mfma_sustained: 256 blocks x
4 waves x 4096 iterations x 16 v_mfma_f32_32x32x16_bf16 and next to nothing else) -- run under
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA -- python tools/pmc_valu_counts_mfma.py
and compare the two counters of mfma_sustained_kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crossclr_amd
from crossclr_amd import _native as nat
lib = nat.library()
out = torch.empty(256 * 256, dtype=torch.float32, device="cuda")
for i in range(3):
    nat.check(lib.crossclr_mfma_sustained(out.data_ptr(), 256, 4096, 100 + i, 0, torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
print("expected MFMA instructions per launch (wave-level):", 256 * 4 * 4096 * 16)
