"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares, the product never routes through the oracle or a CPU fallback, and the Python mirrors keep the
reference's surface (depthwise_conv2d_implicit_gemm.py:11-66, sparse_core.py:80-132)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from slak_amd import build
    return build.build()


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "slak_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(slak_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built_lib):
    syms = _header_symbols()
    assert len(syms) >= 17
    L = ctypes.CDLL(built_lib)
    for s in syms:
        assert hasattr(L, s), "libslak_hip.so does not export %s" % s
    from slak_amd import _lib
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding and header disagree"
    lib = _lib.lib()
    assert lib.slak_version() == 1
    assert lib.slak_status_string(0) == b"ok" and lib.slak_status_string(3) == b"workspace missing or too small"
    # pure host-side argument validation (no GPU needed): status codes instead of exit()
    assert lib.slak_dwconv2d_forward(None, 0, None, 0, None, 0, 1, 1, 1, 1, 3, 3, None, 0, None) == 1
    assert lib.slak_set_conv_algo(99) == 1
    assert lib.slak_dwconv2d_workspace_bytes(0, 128, 96, 56, 56, 51, 5, 2) > 0


def test_library_contains_gfx950_code(built_lib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", built_lib], capture_output=True, text=True).stdout
    if not out:
        pytest.skip("llvm-objdump --offloading unavailable")
    assert "gfx950" in out


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "slak_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "oracle/" in txt and f.endswith((".hip", ".h")):
                    bad.append(f)
    assert not bad, "product files reference the oracle: %s" % bad


def test_no_cpu_fallback_fails_loudly():
    from slak_amd import DepthWiseConv2dImplicitGEMM
    m = DepthWiseConv2dImplicitGEMM(4, (13, 5))
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 4, 8, 8))                        # CPU tensor: refuse, do not silently run F.conv2d
    from slak_amd.sparse_core import CosineDecay, Masking
    import types
    net = nn.Sequential(nn.Linear(8, 8))
    args = types.SimpleNamespace(device="cpu", fix=False, update_frequency=2, only_L=False, sparse_init="uniform", sparsity=0.5, distributed=False)
    from slak_amd._lib import SlakHipError
    with pytest.raises(SlakHipError):
        mk = Masking(torch.optim.SGD(net.parameters(), lr=0.1), None, CosineDecay(0.3, 10), prune_rate=0.3,
                     prune_mode="magnitude", growth_mode="gradient", redistribution_mode="none", args=args)
        mk.add_module(net)                                # apply_mask on CPU params must raise, not fall back
    with pytest.raises(NotImplementedError):
        Masking(None, None, None, growth_mode="random", args=args).init_growth_prune_and_redist()


def test_module_surface_matches_reference():
    from slak_amd import DepthWiseConv2dImplicitGEMM
    import slak_amd.depthwise_conv2d_implicit_gemm as mod
    assert mod.__all__ == ["DepthWiseConv2dImplicitGEMM"]
    m = DepthWiseConv2dImplicitGEMM(6, (51, 5), bias=True)
    assert isinstance(m, nn.Conv2d) and m.groups == 6 and m.padding == (0, 0)
    assert m.weight.shape == (6, 1, 51, 5) and m.bias.shape == (6,)
    ref = nn.Conv2d(6, 6, (51, 5), padding=(25, 2), groups=6, bias=True)
    ref.load_state_dict(m.state_dict())                   # identical state-dict keys / shapes
    import slak_amd.ops as ops
    for n in ("forward_fp32", "backward_data_fp32", "backward_filter_fp32", "forward_fp16", "backward_data_fp16", "backward_filter_fp16"):
        assert callable(getattr(ops, n))                  # frontend.cpp:3-16


def test_cosine_decay_matches_closed_form():
    import math, warnings
    from slak_amd.sparse_core import CosineDecay
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d = CosineDecay(0.3, 50, init_step=3)
        for s in range(4, 10):
            d.step()
            want = 0.005 + (0.3 - 0.005) * (1 + math.cos(math.pi * s / 50)) / 2
            assert abs(d.get_dr(0.3) - want) < 1e-12
