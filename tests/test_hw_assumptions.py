"""Hardware-assumption probes: the MFMA fragment layouts and the ds_read_b64_tr_b16 gather the
kernels (and the CPU emulation shim used by the CPU tests) are written against.  The `gpu` variants
run on the MI355X through crossclr_selftest; the `emu` variants check that the shim models the same
thing, so a CPU-green kernel test means what it says."""
import ctypes
import os

import numpy as np
import pytest
import torch

import crossclr_amd  # noqa: F401
from crossclr_amd import _native as nat

EMU_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "libcrossclr_emu.so")


def _run(which, inp, out_elems, out_dtype, device):
    lib = nat.library()
    tin = torch.from_numpy(inp).to(device)
    tout = torch.zeros(64 * 1024, dtype=torch.uint8, device=device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream if device != "cpu" else 0)
    nat.check(lib.crossclr_selftest(which, ctypes.c_void_p(tin.data_ptr()), ctypes.c_void_p(tout.data_ptr()), stream))
    if device != "cpu":
        torch.cuda.synchronize()
    return tout.cpu().numpy().view(out_dtype)[:out_elems]


def _check_all(device):
    rng = np.random.default_rng(0)
    # (0) v_mfma_f32_32x32x16_bf16 with ASYMMETRIC operands
    a = torch.from_numpy(rng.standard_normal((32, 16)).astype(np.float32)).bfloat16()
    b = torch.from_numpy(rng.standard_normal((16, 32)).astype(np.float32)).bfloat16()
    inp = torch.cat([a.flatten(), b.flatten()]).view(torch.int16).numpy().copy()
    c = _run(0, inp, 1024, np.float32, device).reshape(32, 32)
    ref = (a.float() @ b.float()).numpy()
    assert np.abs(c - ref).max() <= 1e-5, "bf16 32x32x16 fragment layout differs from the documented one"
    # (1) v_mfma_f32_32x32x2_f32
    a = rng.standard_normal((32, 2)).astype(np.float32)
    b = rng.standard_normal((2, 32)).astype(np.float32)
    c = _run(1, np.concatenate([a.ravel(), b.ravel()]), 1024, np.float32, device).reshape(32, 32)
    assert np.abs(c - a @ b).max() <= 1e-6, "f32 32x32x2 fragment layout differs from the documented one"
    # (2) ds_read_b64_tr_b16: lane (n = lane&31, half) must assemble M[8*half + e][n], e = 0..7
    m = np.arange(64 * 64, dtype=np.int16)
    o = _run(2, m, 512, np.int16, device).reshape(64, 8)
    for lane in range(64):
        half, n = lane >> 5, lane & 31
        assert list(o[lane]) == [(8 * half + e) * 64 + n for e in range(8)], f"transpose read, lane {lane}"
    # (4) v_mfma_f32_16x16x32_bf16 (the 16-row backward)
    a = torch.from_numpy(rng.standard_normal((16, 32)).astype(np.float32)).bfloat16()
    b = torch.from_numpy(rng.standard_normal((32, 16)).astype(np.float32)).bfloat16()
    inp = torch.cat([a.flatten(), b.flatten()]).view(torch.int16).numpy().copy()
    c = _run(4, inp, 256, np.float32, device).reshape(16, 16)
    assert np.abs(c - (a.float() @ b.float()).numpy()).max() <= 1e-5, "bf16 16x16x32 fragment layout differs"
    # (3) DPP / swizzle lane exchanges used by the symmetric forward's column-sum butterfly
    v = np.arange(64, dtype=np.float32) + 0.5
    o = _run(3, v, 5 * 64, np.float32, device).reshape(5, 64)
    for k, mask in enumerate((1, 2, 7, 15, 16)):
        assert list(o[k]) == [v[lane ^ mask] for lane in range(64)], f"lane_xor<{mask}>"


@pytest.mark.gpu
def test_fragment_layouts_on_mi355x():
    nat.use_library_for_testing(None)
    assert nat.backend() == "hip-gfx950"
    _check_all("cuda")


def test_fragment_layouts_in_emulation():
    from emu import build_emu
    nat.use_library_for_testing(build_emu.build())
    try:
        assert nat.backend() == "emu-host"
        _check_all("cpu")
    finally:
        nat.use_library_for_testing(None)
