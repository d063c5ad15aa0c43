#!/usr/bin/env python3
"""Compile crossclr_api.cpp for gfx950 with -save-temps and print, for one kernel (regex on the
mangled name), the instruction mix of its main loop and how the LDS waits sit relative to MFMAs.
usage: tools/isa_loop_stats.py 'fast_bwd_kernelILi32' [--dump N]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "crossmodal-contrastive-learning_amd", "csrc", "crossclr_api.cpp")
pat = sys.argv[1]
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-x", "hip", src,
                       "-o", os.path.join(tmp, "x.so"), "-save-temps"] + [a for a in sys.argv[2:] if a.startswith("-D")], cwd=tmp, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "crossclr_api-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
m = re.search(r'^(_ZN8crossclr\w*' + pat + r'\w*):(.*?)\.end_amdhsa_kernel', s, re.S | re.M)
print(m.group(1))
body = m.group(2)
for key in (".vgpr_count", ".agpr_count", ".vgpr_spill_count", "; ScratchSize", "; Occupancy", "; LDSByteSize"):
    mm = re.search(re.escape(key) + r"\S*\s*:?\s*(\S+)", s[m.end():m.end() + 6000] if key.startswith(".") else body)
    if mm: print(" ", key, mm.group(1))
lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]
mf = [i for i, l in enumerate(lines) if l.startswith('v_mfma')]
bars = [i for i, l in enumerate(lines) if l.startswith('s_barrier')]
lo = bars[0] if bars else mf[0]
seg = lines[lo:mf[-1] + 1]
c = collections.Counter(l.split()[0] for l in seg if not l.endswith(':'))
print("loop region:", len(seg), "instructions;", ", ".join(f"{k}={v}" for k, v in c.most_common(18)))
# waits immediately preceding an MFMA
w0 = sum(1 for i in mf if i > 0 and lines[i - 1].startswith('s_waitcnt') and 'lgkmcnt(0)' in lines[i - 1])
wn = sum(1 for i in mf if i > 0 and lines[i - 1].startswith('s_waitcnt') and 'lgkmcnt(0)' not in lines[i - 1])
print(f"MFMAs: {len(mf)}; preceded by lgkmcnt(0): {w0}; by a counted wait: {wn}; scratch ops in loop: {c.get('scratch_load_dword',0)+c.get('scratch_store_dword',0)+c.get('scratch_load_dwordx2',0)+c.get('scratch_load_dwordx4',0)+c.get('scratch_store_dwordx2',0)+c.get('scratch_store_dwordx4',0)}")
if "--dump" in sys.argv:
    n = int(sys.argv[sys.argv.index("--dump") + 1])
    k = mf[len(mf) * 3 // 4]
    print("\n".join("    " + l for l in lines[k - n:k + n]))
