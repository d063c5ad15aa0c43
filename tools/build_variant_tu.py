#!/usr/bin/env python3
"""Tuning variant of ONE leaf translation unit, re-linked against the objects of the product build (seconds instead of the ten minutes
tools/build_variant.py needs for the whole library):
    CROSSCLR_KEEP_OBJS=/tmp/crossclr_objs python crossmodal-contrastive-learning_amd/build.py --force      # once per source state
    python tools/build_variant_tu.py NAME tu_fwdp.cpp [-DFLAG ...]  ->  variants/libNAME.so   (CROSSCLR_HIP_LIBRARY=variants/libNAME.so)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
objs = os.environ.get("CROSSCLR_KEEP_OBJS", "/tmp/crossclr_objs")
sys.path.insert(0, os.path.join(ROOT, "crossmodal-contrastive-learning_amd"))
import build as B
os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
out = os.path.join(ROOT, "variants", f"lib{name}.so")
obj = os.path.join(objs, f"variant_{name}_{unit.replace('.cpp', '.o')}")
subprocess.check_call([B.HIPCC] + B.COMMON + ["-DCROSSCLR_SPLIT", "-x", "hip", "-c", os.path.join(B.CSRC, unit), "-o", obj] + flags)
others = [os.path.join(objs, u.replace(".cpp", ".o")) for u in B.UNITS if u != unit]
missing = [o for o in others if not os.path.exists(o)]
if missing:
    raise SystemExit(f"missing product objects {missing}: run build.py with CROSSCLR_KEEP_OBJS={objs} first")
subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", out])
print(out)
