"""The pipelined kernels keep their LDS fragments in flight through inline-asm loads with hand-counted waits; the host emulation
replaces those helpers with plain loads, so nothing on the CPU side could see a schedule in which hipcc reads (copies, spills) an
asm-loaded register before its wait -- the failure of DESIGN.md section 3.3(3).  This test compiles kernel instantiations for gfx950
(hipcc cross-compiles without a GPU) and runs tools/asm_audit.py over the assembly: no asm-loaded register may be read before a
covering s_waitcnt, none may be pending at a label or branch; and the register-resident kernels must not spill (ScratchSize 0)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crossmodal-contrastive-learning_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

SRC = r'''
#include "crossclr_kernels_fast.h"
namespace crossclr {
#define BSIG (const bf16_t*, const unsigned char*, Geo, const float*, const float*, const float*, const float*, float*, int, int, const float*, const float*)
#define FSIG (const bf16_t*, const bf16_t*, Geo, FwdWork, float*, float*, int*, const float*, const float*, unsigned char*, FwdPerm)
template __global__ void fast_bwd_dsl_kernel<8, false, 0> BSIG;      // local block: bodies M->M, M->D, D->D
template __global__ void fast_bwd_dsl_kernel<8, true, 1> BSIG;       // rectangular (segment walk = real branches), sample weights
template __global__ void fast_bwd_dsl_kernel<8, false, 2> BSIG;      // transposed rectangular (partner gradients)
template __global__ void fast_bwd_dsl_kernel<16, true, 0> BSIG;
template __global__ void fast_bwd_dsl_kernel<32, false, 0> BSIG;     // the headline instantiation
template __global__ void fast_bwd_dsl_kernel<32, false, 0, 2, 4> BSIG;   // two column parts (D = 1024)
template __global__ void fast_bwd_dsl_kernel<32, false, 0, 1, 8, true> BSIG;   // XF: B fragments by asm buffer loads into VGPRs (the headline backward)
template __global__ void fast_bwd_dsl_kernel<8, true, 0, 1, 8, true> BSIG;     // XF, one fragment per wave, sample weights
template __global__ void fast_bwd_dsl_kernel<24, false, 0, 1, 8, true> BSIG;
template __global__ void fast_bwd_dsl_kernel<32, true, 0, 2, 4, true> BSIG;    // XF, two column parts (D = 1024), sample weights
#define PSIG (const unsigned char*, const unsigned char*, unsigned, Geo, const float*, const float*, const float*, const float*, float*, int, int, const float*, const float*)
template __global__ void fast_bwd_xfp_kernel<32, false, 0, 1, 8> PSIG;         // the pair kernel (two tiles per barrier interval): the headline backward
template __global__ void fast_bwd_xfp_kernel<8, true, 0, 1, 8> PSIG;           // one fragment per wave, sample weights
template __global__ void fast_bwd_xfp_kernel<24, false, 0, 1, 8> PSIG;
template __global__ void fast_bwd_xfp_kernel<32, true, 0, 2, 4> PSIG;          // two column parts (D = 1024), sample weights
template __global__ void fast_bwd_xfp_kernel<32, false, 1, 1, 8> PSIG;         // rectangular block (remote columns: reciprocal segment walk)
template __global__ void fast_bwd_xfp_kernel<16, true, 2, 1, 8> PSIG;          // the transpose of a rectangular block (partner gradients)
template __global__ void fast_fwd_pipe_kernel<8, 1, false, true> FSIG;
template __global__ void fast_fwd_pipe_kernel<8, 3, true, true> FSIG;
template __global__ void fast_fwd_pipe_kernel<32, 1, false, true> FSIG;  // the headline forward
template __global__ void fast_fwd_pipe_kernel<64, 2, false, true, 1> FSIG;
#define ZSIG (const bf16_t*, const bf16_t*, Geo, FwdWork, float*, float*, int*, unsigned char*, unsigned, FwdPerm, const float*, const float*)
template __global__ void fast_fwd_pair_kernel<32, true> ZSIG;            // round 5's headline forward (one unbroken MFMA stream per wave)
template __global__ void fast_fwd_pair_kernel<8, false> ZSIG;            // every read of a tile in the double-cadence zone
template __global__ void fast_fwd_pair_kernel<24, true> ZSIG;            // ring of 4 x 24 KiB: wrap by comparison
template __global__ void fast_fwd_pair_kernel<32, true, 1, 2> ZSIG;      // D = 1024: one 32-row half per wave, the tile in two ring stages
template __global__ void fast_fwd_pair_kernel<32, true, 2, 1, 3> ZSIG;   // pairs launch of a sharded run (column sums for the partner, rectangular stash)
template __global__ void fast_fwd_pair_kernel<16, false, 2, 1, 2> ZSIG;  // rectangular launch without column sums
template __global__ void fast_fwd_pair_kernel<32, true, 1, 2, 1, true> ZSIG;   // sample weights (wide operands): the tile's column scales by asm buffer loads (vmcnt-counted)
template __global__ void fast_fwd_pair_kernel<24, true, 1, 2, 3, true> ZSIG;   // ... pairs launch (BASELINE config 5's remote blocks)
}
'''


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_no_asm_loaded_register_is_read_before_its_wait(tmp_path):
    src, asm = tmp_path / "audit.hip", tmp_path / "audit.s"
    src.write_text(SRC)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-x", "hip", "--cuda-device-only", "-S",
                           "-DCROSSCLR_KERNELS_ONLY", "-I", CSRC, str(src), "-o", str(asm)], stderr=subprocess.DEVNULL)
    text = asm.read_text()
    kernels = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", text, re.M)
    assert len(kernels) == 28, kernels
    scratch = [int(x) for x in re.findall(r";\s*ScratchSize:\s*(\d+)", text)]
    assert len(scratch) >= 28 and all(s == 0 for s in scratch), scratch
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm_audit.py"), str(asm)], capture_output=True, text=True)
    assert r.returncode == 0 and "flagged: 0" in r.stdout, r.stdout[-2000:]
    # and the audit itself must be able to see the loads it is meant to guard
    assert text.count("ds_read_b64_tr_b16") > 100 and text.count("ds_read_b128") > 100
    assert len(re.findall(r"buffer_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\], s\d+ offen", text)) > 100
