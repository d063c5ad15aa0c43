#!/usr/bin/env python3
"""Build libcrossclr_hip.so (the C-ABI of include/crossclr.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the source snapshot.

The library is compiled as SEVEN translation units in parallel (-DCROSSCLR_SPLIT): crossclr_api.cpp (the extern "C" boundary, the
generic / projection / exact-fp32 kernels) and one small tu_*.cpp per "leaf" launcher of crossclr_kernels_fast.h, each of which
instantiates one family of the heavy register-resident kernel templates (forward, the three saved backwards, the recomputing
backwards).  Wall time = the longest leaf (~3 minutes) instead of the sum (~10)."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libcrossclr_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc otherwise packs adjacent scalar f32 adds / multiplies of the kernels' epilogues into v_pk_add_f32 / v_pk_mul_f32,
# which cost ~13 cycles more than the two plain VALU they replace beside an MFMA stream (MI355X_MICROARCH.md, "price of one filler"):
# saved backward -2.7 %, forward ~-2 % (profiles/r04_ab_noslp.txt, A/B on one box)
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"]
FLAGS = COMMON + ["-shared", "-x", "hip"]          # (the single-translation-unit form: tools/build_variant.py, tools/isa_loop_stats.py)
UNITS = ["crossclr_api.cpp", "tu_fwd.cpp", "tu_fwdp.cpp", "tu_saved_lds.cpp", "tu_saved_xf1.cpp", "tu_saved_xfp.cpp", "tu_saved_wide.cpp", "tu_recomp.cpp"]


def sources():
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".cpp", ".h"))] + \
           [os.path.join(os.path.dirname(HERE), "include", "crossclr.h")]


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in sources()):
        return OUT
    keep = os.environ.get("CROSSCLR_KEEP_OBJS")      # tools/build_variant_tu.py re-links one re-compiled leaf against these objects
    if keep:
        os.makedirs(keep, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="crossclr_build_") as tmpdir:
        tmp = keep or tmpdir
        procs = []
        for u in UNITS:
            obj = os.path.join(tmp, u.replace(".cpp", ".o"))
            cmd = [HIPCC] + COMMON + ["-DCROSSCLR_SPLIT", "-x", "hip", "-c", os.path.join(CSRC, u), "-o", obj]
            if verbose:
                cmd.append("-Rpass-analysis=kernel-resource-usage")
                print(" ".join(cmd))
            procs.append((u, obj, subprocess.Popen(cmd)))
        failed = [u for u, _, p in procs if p.wait() != 0]
        if failed:
            raise subprocess.CalledProcessError(1, f"hipcc -c {' '.join(failed)}")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for _, obj, _ in procs] + ["-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
