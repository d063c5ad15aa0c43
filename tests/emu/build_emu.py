#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: build tests/emu/libcrossclr_emu.so -- the CrossCLR kernel sources compiled
for the HOST against the lane-level emulation shim (hip_emu.h).  Used only by CPU tests to execute
the real kernel code on tiny shapes; the product library is libcrossclr_hip.so (hipcc, gfx950)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "crossmodal-contrastive-learning_amd", "csrc")
OUT = os.path.join(HERE, "libcrossclr_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".cpp", ".h"))] + \
           [os.path.join(HERE, "hip_emu.h"), os.path.join(HERE, "emu_runtime.cpp")]


def build(force=False):
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in sources()):
        return OUT
    cmd = [CLANG, "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DCROSSCLR_EMU", "-I", HERE, "-I", CSRC,
           "-Wno-unused-value", "-Wno-psabi", os.path.join(CSRC, "crossclr_api.cpp"), os.path.join(HERE, "emu_runtime.cpp"), "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
