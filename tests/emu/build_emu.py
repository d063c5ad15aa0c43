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


UNITS = ["crossclr_api.cpp", "tu_fwd.cpp", "tu_fwdp.cpp", "tu_saved_lds.cpp", "tu_saved_xf1.cpp", "tu_saved_xfp.cpp", "tu_saved_wide.cpp", "tu_recomp.cpp"]


def build(force=False):
    """Seven translation units in parallel, like the product build (crossmodal-contrastive-learning_amd/build.py, -DCROSSCLR_SPLIT)."""
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in sources()):
        return OUT
    import tempfile
    common = [CLANG, "-O2", "-std=c++17", "-fPIC", "-pthread", "-DCROSSCLR_EMU", "-DCROSSCLR_SPLIT", "-I", HERE, "-I", CSRC,
              "-Wno-unused-value", "-Wno-psabi"]
    with tempfile.TemporaryDirectory(prefix="crossclr_emu_") as tmp:
        procs = []
        for u in UNITS + ["emu_runtime.cpp"]:
            src = os.path.join(HERE if u == "emu_runtime.cpp" else CSRC, u)
            obj = os.path.join(tmp, u.replace(".cpp", ".o"))
            procs.append((u, obj, subprocess.Popen(common + ["-c", src, "-o", obj])))
        failed = [u for u, _, p in procs if p.wait() != 0]
        if failed:
            raise subprocess.CalledProcessError(1, "clang++ -c " + " ".join(failed))
        subprocess.check_call([CLANG, "-shared", "-fPIC", "-pthread"] + [obj for _, obj, _ in procs] + ["-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
