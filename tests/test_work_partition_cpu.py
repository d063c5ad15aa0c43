"""The persistent forward's work partition (FwdWork in csrc/crossclr_device.h: ranges of equal COST over the flat (row block, tile)
list) is pure integer code shared by the forward kernels, the plan (fwd_slots) and fwd_finish_kernel (which slots to add up).
tests/native/fwd_partition_check.cpp exercises it on the host for every padded batch up to 2048 rows, several block counts, both row-block
heights and the three list kinds: exact tiling, owner blocks == first..last block of every row block, slot bound, cost balance."""
import os
import subprocess

import pytest

from tests.emu import build_emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_forward_work_partition_invariants(tmp_path):
    cxx = build_emu.CLANG
    if not os.path.exists(cxx):
        pytest.skip("the host clang++ of the ROCm image is not installed here")
    exe = str(tmp_path / "fwd_partition_check")
    subprocess.check_call([cxx, "-O1", "-std=c++17", "-DCROSSCLR_EMU", "-I", os.path.join(ROOT, "tests", "emu"),
                           "-I", os.path.join(ROOT, "crossmodal-contrastive-learning_amd", "csrc"), "-Wno-psabi", "-Wno-unused-value",
                           os.path.join(ROOT, "tests", "native", "fwd_partition_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "partition ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
