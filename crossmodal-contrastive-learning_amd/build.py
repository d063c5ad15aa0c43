#!/usr/bin/env python3
"""Build libcrossclr_hip.so (the C-ABI of include/crossclr.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the source snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcrossclr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip"]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".cpp", ".h"))] + \
           [os.path.join(os.path.dirname(HERE), "include", "crossclr.h")]


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in sources()):
        return OUT
    cmd = [HIPCC] + FLAGS + [os.path.join(CSRC, "crossclr_api.cpp"), "-o", OUT]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
