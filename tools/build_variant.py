#!/usr/bin/env python3
"""Build a tuning variant of the HIP library: tools/build_variant.py NAME [-DFLAG ...] -> variants/libNAME.so
(select it at run time with CROSSCLR_HIP_LIBRARY=variants/libNAME.so)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, flags = sys.argv[1], sys.argv[2:]
os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
out = os.path.join(ROOT, "variants", f"lib{name}.so")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "-fno-slp-vectorize",
                       os.path.join(ROOT, "crossmodal-contrastive-learning_amd", "csrc", "crossclr_api.cpp"), "-o", out] + flags)
print(out)
