"""fp32 tensors on the bf16 matrix cores (pytest -m gpu): the opt-in two-term split of include/slak_hip.h `slak_set_fp32_matrix_cores`.

x = bf16(x) + bf16(x - bf16(x)) keeps 16 significand bits per operand; every product is three MFMAs (hi*hi + hi*lo + lo*hi) into an fp32
accumulator.  The error of an output is therefore bounded by ~3 * 2^-17 * sum|x||w| (two representation errors and the dropped lo*lo term);
the tests assert 2^-15 * sum|x||w| against the C oracle (fp64 accumulate), i.e. >= 30x inside the 1e-3 of the reference's own test
(test_correctness.py) on these inputs, and that the default (switch off) stays on the exact kernels.
"""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

SHAPES = [(3, 5, 56, 56, 51, 5), (3, 5, 56, 56, 5, 51), (2, 3, 56, 56, 5, 5), (4, 6, 28, 28, 49, 5), (4, 6, 28, 28, 5, 49), (5, 4, 14, 14, 47, 5),
          (5, 4, 14, 14, 5, 47), (6, 8, 7, 7, 13, 5), (6, 8, 7, 7, 5, 13), (2, 3, 48, 40, 5, 31), (2, 3, 40, 48, 31, 5), (9, 2, 24, 24, 57, 5)]


@pytest.fixture
def split_on():
    from slak_amd import ops
    prev = ops.allow_fp32_matrix_cores(True)
    yield
    ops.allow_fp32_matrix_cores(prev)


def _last():
    from slak_amd import _lib
    return _lib.lib().slak_debug_last_kernel().decode()


@pytest.mark.parametrize("N,C,H,W,kh,kw", SHAPES)
def test_fp32_split_forward_backward_vs_oracle(N, C, H, W, kh, kw, gpu, split_on):
    from slak_amd import ops
    rng = np.random.default_rng(H * 100 + kh)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    dy = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((C, 1, kh, kw)) * 0.05).astype(np.float32)
    xg, dyg, wg = (torch.from_numpy(a).to(gpu) for a in (x, dy, w))
    eps = 2.0 ** -15
    y = ops.dwconv2d_forward(xg, wg)
    assert _last() == "dwconv_mfma(f32 split)" and y.dtype == torch.float32
    bound = eps * oracle.dwconv2d_fwd(np.abs(x), np.abs(w)) + 1e-30
    err = np.abs(y.cpu().numpy().astype(np.float64) - oracle.dwconv2d_fwd(x, w))
    assert (err <= bound).all(), float((err / bound).max())
    dx = ops.dwconv2d_backward_data(dyg, wg)
    assert _last() == "dwconv_mfma(f32 split)"
    bound = eps * oracle.dwconv2d_bwd_data(np.abs(dy), np.abs(w)) + 1e-30
    err = np.abs(dx.cpu().numpy().astype(np.float64) - oracle.dwconv2d_bwd_data(dy, w))
    assert (err <= bound).all(), float((err / bound).max())
    dw = ops.dwconv2d_backward_filter(dyg, xg, wg)
    assert _last() == "dwconv_mfma_wgrad(f32 split)" and dw.dtype == torch.float32
    bound = eps * oracle.dwconv2d_bwd_filter(np.abs(dy), np.abs(x), kh, kw) + 1e-30
    err = np.abs(dw.cpu().numpy().astype(np.float64) - oracle.dwconv2d_bwd_filter(dy, x, kh, kw))
    assert (err <= bound).all(), float((err / bound).max())
    assert torch.equal(dw, ops.dwconv2d_backward_filter(dyg, xg, wg))             # fixed-order reduction
    # the reference test's own criterion (test_correctness.py: 1e-3 of the result) with a wide margin on the values that are not cancellations
    ref = oracle.dwconv2d_fwd(x, w)
    big = np.abs(ref) > 0.1 * np.abs(ref).max()
    assert (np.abs(y.cpu().numpy() - ref)[big] <= 1e-4 * np.abs(ref)[big]).all()


def test_fp32_stays_exact_unless_allowed(gpu):
    from slak_amd import _lib, ops
    assert _lib.lib().slak_get_fp32_matrix_cores() == 0
    x = torch.randn(2, 3, 28, 28, device=gpu); w = torch.zeros(3, 1, 49, 5, device=gpu); w[:, 0, 24, 2] = 1
    y = ops.dwconv2d_forward(x, w)
    assert _last() == "dwconv_direct" and torch.equal(y, x)                       # a delta filter is the identity, bit for bit
    prev = ops.allow_fp32_matrix_cores(True)
    try:
        y2 = ops.dwconv2d_forward(x, w)
        assert _last() == "dwconv_mfma(f32 split)"
        assert (y2 - x).abs().max().item() <= 2.0 ** -16 * x.abs().max().item()  # 16 significand bits survive the split
        assert not torch.equal(y2, x)
    finally:
        ops.allow_fp32_matrix_cores(prev)
    # a non-finite value stays inside its own plane (either path turns the taps around it into inf * 0 = NaN)
    x[0, 0, 3, 3] = float("inf")
    prev = ops.allow_fp32_matrix_cores(True)
    try:
        y3 = ops.dwconv2d_forward(x, w)
        assert not torch.isfinite(y3[0, 0, 3, 3]) and torch.isfinite(y3[0, 1:]).all() and torch.isfinite(y3[1]).all()
    finally:
        ops.allow_fp32_matrix_cores(prev)


def test_fp32_under_autocast_is_exact_by_default_and_takes_the_matrix_cores_when_opted_in(gpu):
    """The op module's policy (slak_amd/depthwise_conv2d_implicit_gemm.py): by default an fp32 activation runs the exact kernels, in and out of
    torch.autocast, and follows the process-wide switch.  With the class switch ON (opt-in), an fp32 activation entered under
    torch.autocast -- the reference's default AMP flow, depthwise_conv2d_implicit_gemm.py:16 -- takes the two-term split in forward, data
    gradient and weight gradient; results stay fp32 either way."""
    from slak_amd import _lib
    from slak_amd.depthwise_conv2d_implicit_gemm import DepthWiseConv2dImplicitGEMM
    L = _lib.lib()
    torch.manual_seed(3)
    m = DepthWiseConv2dImplicitGEMM(6, (31, 5)).to(gpu)
    x = torch.randn(3, 6, 28, 28, device=gpu, requires_grad=True)
    y0 = m(x)
    assert L.slak_debug_last_kernel() == b"dwconv_direct" and y0.dtype == torch.float32
    assert DepthWiseConv2dImplicitGEMM.fp32_matrix_cores_under_autocast is False            # the default: opt-in
    with torch.autocast("cuda", dtype=torch.bfloat16):
        yd = m(x)
    assert L.slak_debug_last_kernel() == b"dwconv_direct" and torch.equal(yd, y0)
    from slak_amd import ops
    prev = ops.allow_fp32_matrix_cores(True)                                               # the process-wide switch is followed ...
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            m(x)
        assert L.slak_debug_last_kernel() == b"dwconv_mfma(f32 split)"
    finally:
        ops.allow_fp32_matrix_cores(prev)
    with torch.autocast("cuda", dtype=torch.bfloat16):                                     # ... and restoring it restores exact fp32
        assert torch.equal(m(x), y0)
    DepthWiseConv2dImplicitGEMM.fp32_matrix_cores_under_autocast = True
    try:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y1 = m(x)
    finally:
        DepthWiseConv2dImplicitGEMM.fp32_matrix_cores_under_autocast = False
    assert L.slak_debug_last_kernel() == b"dwconv_mfma(f32 split)" and y1.dtype == torch.float32
    assert (y1 - y0).abs().max().item() <= 1e-4 * y0.abs().max().item()
    dy = torch.randn_like(y0)
    gx0, gw0 = torch.autograd.grad(y0, (x, m.weight), dy)                        # exact
    gx1, gw1 = torch.autograd.grad(y1, (x, m.weight), dy)                        # the backward runs on autograd's thread: the override travels in ctx
    assert gx1.dtype == torch.float32 and gw1.dtype == torch.float32
    for a, b in ((gx1, gx0), (gw1, gw0)):
        assert not torch.equal(a, b) and (a - b).abs().max().item() <= 1e-4 * b.abs().max().item()
    assert L.slak_get_fp32_matrix_cores_effective() == 0                          # the per-thread override did not leak
