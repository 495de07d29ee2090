"""The compiled pybind boundary: `_depthwise_conv2d_implicit_gemm_C` (slak_amd/pybind/frontend_hip.cpp on libslak_hip.so) exports
what the reference's frontend.cpp:3-16 exports, so the reference's own depthwise_conv2d_implicit_gemm.py imports on it unmodified."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

REF_EXT_DIR = "/root/reference/cutlass/examples/19_large_depthwise_conv2d_torch_extension"
NAMES = ["forward_fp32", "backward_data_fp32", "backward_filter_fp32", "forward_fp16", "backward_data_fp16", "backward_filter_fp16"]


def _ext():
    from slak_amd import build
    path = build.build_pybind()
    d = os.path.dirname(path)
    if d not in sys.path:
        sys.path.insert(0, d)
    return importlib.import_module(build.PYBIND_NAME)


def test_pybind_module_builds_and_exports_the_reference_names():
    E = _ext()
    for n in NAMES + ["forward_bf16", "backward_data_bf16", "backward_filter_bf16"]:
        assert callable(getattr(E, n)), n
    with pytest.raises(RuntimeError):                     # TORCH_CHECK instead of exit(): CPU tensors are refused (forward_fp32.cu:194-196)
        E.forward_fp32(torch.zeros(1, 1, 4, 4), torch.zeros(1, 1, 3, 3))


@pytest.mark.skipif(not os.path.isdir(REF_EXT_DIR), reason="reference checkout not present (build container only)")
def test_reference_module_file_imports_on_it_unmodified():
    _ext()
    sys.path.insert(0, REF_EXT_DIR)
    try:
        sys.modules.pop("depthwise_conv2d_implicit_gemm", None)
        ref = importlib.import_module("depthwise_conv2d_implicit_gemm")
        assert ref.__file__.startswith(REF_EXT_DIR)
        m = ref.DepthWiseConv2dImplicitGEMM(6, (51, 5), bias=False)
        assert isinstance(m, torch.nn.Conv2d) and tuple(m.weight.shape) == (6, 1, 51, 5)
    finally:
        sys.path.remove(REF_EXT_DIR)
        sys.modules.pop("depthwise_conv2d_implicit_gemm", None)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,sfx", [(torch.float32, "fp32"), (torch.float16, "fp16"), (torch.bfloat16, "bf16")])
@pytest.mark.parametrize("N,C,H,W,kh,kw", [(3, 4, 28, 28, 49, 5), (2, 3, 14, 14, 5, 47), (2, 5, 9, 11, 7, 7)])
def test_pybind_entry_points_match_the_ctypes_path_and_the_oracle(N, C, H, W, kh, kw, dtype, sfx, gpu):
    import oracle
    from slak_amd import ops
    E = _ext()
    torch.manual_seed(kh + H)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dy = torch.randn(N, C, H, W, device=gpu).to(dtype)
    w = torch.randn(C, 1, kh, kw, device=gpu) * 0.05
    y = getattr(E, "forward_" + sfx)(x, w)
    dx = getattr(E, "backward_data_" + sfx)(dy, w)
    dw = getattr(E, "backward_filter_" + sfx)(dy, x, w)
    assert y.dtype == dtype and dx.dtype == dtype and dw.dtype == torch.float32 and tuple(dw.shape) == (C, 1, kh, kw)
    assert torch.equal(y, ops.dwconv2d_forward(x, w)) and torch.equal(dx, ops.dwconv2d_backward_data(dy, w))
    assert torch.equal(dw, ops.dwconv2d_backward_filter(dy, x, w))
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    ref = oracle.dwconv2d_fwd(x.float().cpu().numpy(), w.cpu().numpy())
    assert np.abs(y.double().cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max())
    with pytest.raises(RuntimeError):
        E.forward_fp32(x.transpose(2, 3), w)              # non-contiguous (forward_fp32.cu:203-204)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):                            # the CURRENT stream, not the null stream (convolution.h:243)
        y2 = getattr(E, "forward_" + sfx)(x, w)
    s.synchronize()
    assert torch.equal(y, y2)


@pytest.mark.gpu
def test_reference_module_runs_on_it(gpu):
    """The reference's own depthwise_conv2d_implicit_gemm.py on the compiled module: from /root/reference where it exists, from the
    bytecode oracle/ref_modules.build() left under oracle/_ref/ on the GPU box (more shapes: tests/test_reference_modules_gpu.py)."""
    from oracle import ref_modules
    _ext()
    from slak_amd import build
    ref = ref_modules.load_dwconv_module(os.path.dirname(build.pybind_path()))
    m = ref.DepthWiseConv2dImplicitGEMM(6, (51, 5)).to(gpu)
    x = torch.randn(2, 6, 28, 28, device=gpu, requires_grad=True)
    y = m(x); y.mean().backward()
    want = torch.nn.functional.conv2d(x.detach().cpu().double(), m.weight.detach().cpu().double(), None, 1, (25, 2), 1, 6)
    assert (y.detach().cpu().double() - want).abs().max().item() <= 1e-4
