#!/usr/bin/env python3
"""Compile ONE kernel instantiation of crossclr_kernels_fast.h to gfx950 assembly (seconds, not a minute) and summarise it.
usage: tools/kernel_asm.py 'fast_bwd_saved_kernel<32, false>' 'const bf16_t*, const unsigned char*, Geo, const float*, const float*, float*, int, int, const float*' [--loop] [-DFLAG...]
Prints registers / scratch / occupancy, every control-flow or wait instruction with its index, and (--loop) the loop body."""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crossmodal-contrastive-learning_amd", "csrc")
kern, sig = sys.argv[1], sys.argv[2]
flags = [a for a in sys.argv[3:] if a.startswith("-D")]
tmp = tempfile.mkdtemp()
src = os.path.join(tmp, "k.hip")
open(src, "w").write('#include "crossclr_kernels_fast.h"\nnamespace crossclr {\ntemplate __global__ void %s(%s);\n}\n' % (kern, sig))
out = os.path.join(tmp, "k.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S",
                       "-DCROSSCLR_KERNELS_ONLY", "-I", CSRC, src, "-o", out] + flags + [a for a in sys.argv[3:] if a.startswith(("-f", "-m"))], stderr=subprocess.DEVNULL)
s = open(out).read()
for key in ("; NumVgprs", "; NumAgprs", "; ScratchSize", "; Occupancy", "; LDSByteSize", "; TotalNumSgprs"):
    m = re.search(re.escape(key) + r":\s*(\S+)", s)
    if m: print(key[2:], m.group(1), end="  ")
print()
lines = [l.strip() for l in s.split("\n") if l.strip() and not l.strip().startswith((".", ";"))]
mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
print("instructions", len(lines), "mfma", len(mf), "first/last", mf[0], mf[-1])
lo, hi = max(0, mf[0] - 400), min(len(lines), mf[-1] + 400)
for i in range(lo, hi):
    l = lines[i]
    if l.endswith(":") or l.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm", "scratch_")) or "vmcnt" in l:
        print(i, l)
seg = lines[lo:hi]
c = collections.Counter(l.split()[0] for l in seg if not l.endswith(":"))
print("mix:", ", ".join(f"{k}={v}" for k, v in c.most_common(30)))
if "--loop" in sys.argv:
    print("\n".join(f"{i}: {lines[i]}" for i in range(lo, hi)))
print("asm:", out)
