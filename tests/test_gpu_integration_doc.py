"""INTEGRATION.md shows the ctypes binding a maintainer of the reference would add inside its own `forward`.  This test runs
that very code block (extracted from the document) on the MI355X and compares it with the module."""
import os
import re

import pytest
import torch

import crossclr_amd
from crossclr_amd import _native as nat

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_documented_ctypes_stub_runs_and_matches_the_module():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"```python\nimport ctypes, torch\n(.*?)```", text, re.S)
    assert m, "the binding example is missing from INTEGRATION.md"
    code = "import ctypes, torch\n" + m.group(1).replace('ctypes.CDLL("libcrossclr_hip.so")', f'ctypes.CDLL("{nat.library_path()}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    g = torch.Generator().manual_seed(0)
    v, t = torch.randn(1000, 300, generator=g).cuda(), torch.randn(1000, 300, generator=g).cuda()
    got = ns["crossclr_forward_only"](v, t)
    with torch.no_grad():
        want = crossclr_amd.CrossCLR_onlyIntraModality(0.03, 0.8, compute_mode="bf16").cuda()(v, t)
    assert got.dtype == torch.float64 and got.dim() == 0
    assert abs(got.item() - want.item()) <= 1e-9
