"""GPU parity tests of the depthwise-conv hot path (run on the MI355X box: pytest -m gpu).

Every call goes  DepthWiseConv2dImplicitGEMM / slak_amd.ops  ->  ctypes  ->  C ABI (include/slak_hip.h)
->  HIP kernels.  The checker is the oracle (oracle/, CPU) and the committed golden fixtures made from
the reference's own ground truth F.conv2d (tests/golden/make_golden.py).  Tolerances (BASELINE.json):
fp32 within 1e-3 (we assert much tighter: 1e-4 rel / 1e-5 abs), bf16/fp16 within 1e-2.
"""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import oracle
from conftest import golden_conv_cases, load_golden

pytestmark = pytest.mark.gpu

FP32_RTOL, FP32_ATOL = 1e-4, 1e-5        # north star: 1e-3
LOWP_TOL = 1e-2                          # north star: 1e-2 (bf16); also used for fp16


def _ops():
    from slak_amd import ops
    return ops


def _assert_close(got, ref64, rtol, atol, what):
    got = got.detach().double().cpu().numpy()
    err = np.abs(got - ref64)
    bound = atol + rtol * np.abs(ref64)
    worst = float((err - bound).max())
    assert worst <= 0, "%s: max abs err %.3e (|ref|max %.3e), violates atol=%g rtol=%g by %.3e" % (
        what, err.max(), np.abs(ref64).max(), atol, rtol, worst)


# ------------------------------------------------------------------ golden fixtures, fp32, all three passes
@pytest.mark.parametrize("case", golden_conv_cases())
def test_golden_fp32(case, gpu):
    g = load_golden("dwconv_" + case)
    ops = _ops()
    x, w, dy = (torch.from_numpy(g[k]).to(gpu) for k in ("x", "w", "dy"))
    _assert_close(ops.dwconv2d_forward(x, w), g["y64"], FP32_RTOL, FP32_ATOL, "fwd " + case)
    _assert_close(ops.dwconv2d_backward_data(dy, w), g["dx64"], FP32_RTOL, FP32_ATOL, "dgrad " + case)
    N = x.shape[0] * x.shape[2] * x.shape[3]
    _assert_close(ops.dwconv2d_backward_filter(dy, x, w), g["dw64"], FP32_RTOL, FP32_ATOL * max(1.0, N ** 0.5), "wgrad " + case)


@pytest.mark.parametrize("case", ["lora1_51x5_s1", "lora2_5x51_s1", "small_5x5_s1", "lora1_49x5_s2", "lora2_5x47_s3", "lora1_13x5_s4", "ragged_9x11"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_golden_lowp(case, dtype, gpu):
    """bf16/fp16 I/O, fp32 accumulate: compare with the oracle evaluated on the ROUNDED inputs."""
    g = load_golden("dwconv_" + case)
    ops = _ops()
    x, dy = (torch.from_numpy(g[k]).to(gpu).to(dtype) for k in ("x", "dy"))
    w = torch.from_numpy(g["w"]).to(gpu)                      # fp32 master weights, read directly
    xr, dyr = x.float().cpu().numpy(), dy.float().cpu().numpy()
    kh, kw = w.shape[2:]
    y = ops.dwconv2d_forward(x, w)
    assert y.dtype == dtype
    yref = oracle.dwconv2d_fwd(xr, g["w"])
    scale = max(1.0, float(np.abs(yref).max()))
    _assert_close(y, yref, LOWP_TOL, LOWP_TOL * scale, "fwd")
    dx = ops.dwconv2d_backward_data(dy, w)
    assert dx.dtype == dtype
    dxref = oracle.dwconv2d_bwd_data(dyr, g["w"])
    _assert_close(dx, dxref, LOWP_TOL, LOWP_TOL * max(1.0, float(np.abs(dxref).max())), "dgrad")
    dw = ops.dwconv2d_backward_filter(dy, x, w)
    assert dw.dtype == torch.float32                           # backward_filter_fp16.cu:187
    dwref = oracle.dwconv2d_bwd_filter(dyr, xr, kh, kw)
    _assert_close(dw, dwref, LOWP_TOL, LOWP_TOL * max(1.0, float(np.abs(dwref).max())), "wgrad")


# ------------------------------------------------------------------ the reference's own test grid
# test_correctness.py:15-35 (fwd fp32) and :61-90 (bwd fp32): same shapes, seeds, randn inputs, default
# Conv2d init, y.mean().backward().  Reference tolerance is allclose default; ours is stated above.
@pytest.mark.parametrize("batch_size", [1, 16])
@pytest.mark.parametrize("channels", [64, 192])
@pytest.mark.parametrize("kernel_size", [3, 7, 13, 31])
@pytest.mark.parametrize("resolution", [16, 32])
@pytest.mark.parametrize("seed", [0, 42])
def test_forward_fp32_reference_grid(batch_size, channels, kernel_size, resolution, seed, gpu):
    from slak_amd import DepthWiseConv2dImplicitGEMM
    torch.random.manual_seed(seed)
    x = torch.randn(batch_size, channels, resolution, resolution)
    m = DepthWiseConv2dImplicitGEMM(channels, kernel_size)
    y_ref = F.conv2d(x.double(), m.weight.detach().double(), padding=kernel_size // 2, groups=channels)
    y = m.to(gpu)(x.to(gpu))
    assert y.dtype == torch.float
    _assert_close(y, y_ref.numpy(), FP32_RTOL, FP32_ATOL, "fwd grid")


@pytest.mark.parametrize("batch_size", [1, 16])
@pytest.mark.parametrize("channels", [64])
@pytest.mark.parametrize("kernel_size", [3, 7, 13])
@pytest.mark.parametrize("resolution", [16])
@pytest.mark.parametrize("seed", [0, 42])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_backward_reference_grid(batch_size, channels, kernel_size, resolution, seed, dtype, gpu):
    from slak_amd import DepthWiseConv2dImplicitGEMM
    torch.random.manual_seed(seed)
    x = torch.randn(batch_size, channels, resolution, resolution)
    m = DepthWiseConv2dImplicitGEMM(channels, kernel_size)
    xr = x.to(dtype).double().requires_grad_(True)
    wr = m.weight.detach().double().requires_grad_(True)
    F.conv2d(xr, wr, padding=kernel_size // 2, groups=channels).mean().backward()
    xg = x.to(gpu).to(dtype).requires_grad_(True)
    m = m.to(gpu)
    if dtype == torch.float32:
        y = m(xg)
    else:
        with torch.autocast("cuda", dtype=dtype):
            y = m(xg)
    assert y.dtype == dtype
    y.mean().backward()
    assert xg.grad.dtype == dtype and m.weight.grad.dtype == torch.float32     # test_correctness.py:121-124
    if dtype == torch.float32:
        _assert_close(xg.grad, xr.grad.numpy(), FP32_RTOL, 1e-8, "dx grid")
        _assert_close(m.weight.grad, wr.grad.numpy(), 1e-4, 1e-6, "dw grid")      # test_correctness.py:90
    else:
        _assert_close(xg.grad, xr.grad.numpy(), LOWP_TOL, 1e-6, "dx grid")
        _assert_close(m.weight.grad, wr.grad.numpy(), LOWP_TOL, 1e-4, "dw grid")


# ------------------------------------------------------------------ module surface (drop-in contract)
def test_module_is_conv2d_and_loads_conv2d_state(gpu):
    from slak_amd import DepthWiseConv2dImplicitGEMM
    torch.manual_seed(0)
    m1 = DepthWiseConv2dImplicitGEMM(12, (13, 5), bias=True).to(gpu)
    m2 = nn.Conv2d(12, 12, (13, 5), padding=(6, 2), groups=12, bias=True).to(gpu)
    m2.load_state_dict(m1.state_dict())                          # depthwise_conv2d_implicit_gemm.py:73-75
    assert isinstance(m1, nn.Conv2d) and m1.padding == (0, 0) and m1.weight.shape == (12, 1, 13, 5)
    x = torch.randn(3, 12, 20, 17, device=gpu)
    torch.testing.assert_close(m1(x), m2(x), rtol=1e-4, atol=1e-5)


def test_int_kernel_and_tuple_kernel(gpu):
    from slak_amd import DepthWiseConv2dImplicitGEMM
    assert DepthWiseConv2dImplicitGEMM(4, 5).weight.shape == (4, 1, 5, 5)
    assert DepthWiseConv2dImplicitGEMM(4, (5, 51)).weight.shape == (4, 1, 5, 51)


def test_error_behaviour(gpu):
    from slak_amd import DepthWiseConv2dImplicitGEMM, ops
    from slak_amd._lib import SlakHipError
    m = DepthWiseConv2dImplicitGEMM(4, 5).to(gpu)
    with pytest.raises(TypeError):                                # depthwise_conv2d_implicit_gemm.py:63
        m(torch.randn(1, 4, 8, 8, device=gpu, dtype=torch.float64))
    with pytest.raises(RuntimeError):                             # forward_fp32.cu:194-196
        ops.dwconv2d_forward(torch.randn(1, 4, 8, 8), m.weight)
    with pytest.raises(RuntimeError):                             # forward_fp32.cu:203-204 (contiguity)
        ops.dwconv2d_forward(torch.randn(1, 4, 8, 16, device=gpu)[..., ::2], m.weight)
    with pytest.raises(SlakHipError):                             # even kernels are rejected, not mis-sized
        ops.dwconv2d_forward(torch.randn(1, 4, 8, 8, device=gpu), torch.randn(4, 1, 4, 4, device=gpu))


def test_autocast_dtype_flow(gpu):
    """fp32 input under autocast stays on the fp32 path (custom_fwd(cast_inputs=float32),
    depthwise_conv2d_implicit_gemm.py:16); bf16 input takes the bf16 path."""
    from slak_amd import DepthWiseConv2dImplicitGEMM
    m = DepthWiseConv2dImplicitGEMM(8, (13, 5)).to(gpu)
    x = torch.randn(2, 8, 14, 14, device=gpu)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert m(x).dtype == torch.float32
        assert m(x.bfloat16()).dtype == torch.bfloat16


def test_non_default_stream(gpu):
    ops = _ops()
    g = load_golden("dwconv_lora1_13x5_s4")
    x, w = torch.from_numpy(g["x"]).to(gpu), torch.from_numpy(g["w"]).to(gpu)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        y = ops.dwconv2d_forward(x, w)
    s.synchronize()
    _assert_close(y, g["y64"], FP32_RTOL, FP32_ATOL, "stream")


# ------------------------------------------------------------------ edge shapes
@pytest.mark.parametrize("N,C,H,W,kh,kw", [
    (1, 1, 1, 1, 3, 3), (3, 5, 1, 9, 5, 7), (3, 5, 9, 1, 7, 5), (5, 3, 7, 7, 51, 5), (5, 3, 7, 7, 5, 51),
    (2, 130, 6, 5, 3, 3), (7, 2, 33, 31, 13, 5), (1, 2, 96, 96, 61, 5), (1, 2, 96, 96, 5, 61), (4, 3, 24, 24, 5, 57),
])
def test_edge_shapes_vs_oracle(N, C, H, W, kh, kw, gpu):
    ops = _ops()
    rng = np.random.default_rng(N * 1000 + H * 10 + kh)
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    dy = rng.standard_normal((N, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((C, 1, kh, kw)) * 0.05).astype(np.float32)
    xg, dyg, wg = (torch.from_numpy(a).to(gpu) for a in (x, dy, w))
    _assert_close(ops.dwconv2d_forward(xg, wg), oracle.dwconv2d_fwd(x, w), FP32_RTOL, FP32_ATOL, "fwd")
    _assert_close(ops.dwconv2d_backward_data(dyg, wg), oracle.dwconv2d_bwd_data(dy, w), FP32_RTOL, FP32_ATOL, "dgrad")
    _assert_close(ops.dwconv2d_backward_filter(dyg, xg, wg), oracle.dwconv2d_bwd_filter(dy, x, kh, kw), FP32_RTOL, 1e-4, "wgrad")


# ------------------------------------------------------------------ BASELINE.json full sizes: size-independent properties
FULL = [  # cfg 2/3 per-GPU shapes and the cfg-5 stage-1 shape at reduced batch
    (128, 96, 56, 56, 51, 5), (128, 96, 56, 56, 5, 51), (128, 96, 56, 56, 5, 5),
    (128, 192, 28, 28, 49, 5), (128, 384, 14, 14, 5, 47), (128, 768, 7, 7, 13, 5),
    (16, 96, 96, 96, 61, 5),
]


@pytest.mark.parametrize("N,C,H,W,kh,kw", FULL)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_full_size_properties(N, C, H, W, kh, kw, dtype, gpu):
    """(1) adjointness  <conv(x,w),dy> == <x,dgrad(dy,w)> == <w,wgrad(dy,x)>  (fp64 dot products);
       (2) linearity in w;  (3) a delta filter is the identity;  (4) wgrad is bitwise reproducible;
       (5) a random sample of planes agrees with the oracle."""
    ops = _ops()
    torch.manual_seed(kh * 100 + H)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dy = torch.randn(N, C, H, W, device=gpu).to(dtype)
    w = (torch.randn(C, 1, kh, kw, device=gpu) * 0.02)
    y = ops.dwconv2d_forward(x, w)
    dx = ops.dwconv2d_backward_data(dy, w)
    dw = ops.dwconv2d_backward_filter(dy, x, w)
    a = (y.double() * dy.double()).sum().item()
    b = (dx.double() * x.double()).sum().item()
    c = (dw.double() * w.double()).sum().item()
    norm = (y.double().norm() * dy.double().norm()).item()
    tol = 1e-6 if dtype == torch.float32 else 2e-3            # low precision: y and dx are rounded to bf16
    assert abs(a - b) <= tol * norm and abs(a - c) <= tol * norm, (a, b, c, norm)
    # linearity in the filter
    w2 = torch.randn_like(w) * 0.02
    y12 = ops.dwconv2d_forward(x, w + w2).float()
    y1p2 = y.float() + ops.dwconv2d_forward(x, w2).float()
    lin_tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (y12 - y1p2).abs().max().item() <= lin_tol * max(1.0, y12.abs().max().item())
    # identity filter
    wi = torch.zeros_like(w); wi[:, 0, kh // 2, kw // 2] = 1
    assert torch.equal(ops.dwconv2d_forward(x, wi), x)
    # determinism of the weight gradient (no atomics)
    assert torch.equal(dw, ops.dwconv2d_backward_filter(dy, x, w))
    # sampled planes vs oracle
    ns = [0, N // 2, N - 1]; cs = [0, C // 3, C - 1]
    xs = x[ns][:, cs].float().cpu().numpy(); ws = w[cs].cpu().numpy()
    yref = oracle.dwconv2d_fwd(xs, ws)
    if dtype == torch.float32:
        _assert_close(y[ns][:, cs], yref, FP32_RTOL, FP32_ATOL, "sample fwd")
    else:
        _assert_close(y[ns][:, cs], yref, LOWP_TOL, LOWP_TOL * max(1.0, float(np.abs(yref).max())), "sample fwd")
