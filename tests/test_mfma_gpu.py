"""GPU parity tests of the matrix-core (MFMA) depthwise-conv kernels (pytest -m gpu, MI355X).

All calls go through the C ABI (slak_amd.ops -> ctypes -> libslak_hip.so).  The checker is the CPU oracle on the
ROUNDED operands: the MFMA path rounds the fp32 filter to the activation dtype (what autocast does to an nn.Conv2d
weight), accumulates in fp32 and rounds the result once, so against an fp64 evaluation of the same rounded operands the
only error is the output rounding (half an ulp: 2^-8 relative for bf16, 2^-11 for fp16) plus fp32 accumulation noise.
Tolerance asserted: 1e-2 (BASELINE.json, bf16) -- and the tighter "half an output ulp + accumulation" bound below.
"""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

LOWP_TOL = 1e-2


def _ops():
    from slak_amd import ops
    return ops


def _lib():
    from slak_amd import _lib
    return _lib


@pytest.fixture()
def mfma_only(gpu):
    """Force the MFMA algorithm (unsupported shapes raise instead of silently running the direct kernels)."""
    L = _lib()
    L.lib().slak_set_conv_algo(L.ALGO_MFMA)
    yield
    L.lib().slak_set_conv_algo(L.ALGO_AUTO)


def _round(t, dtype):
    return t.to(dtype).float().cpu().numpy()


def _check(got, ref, ulp_rel, what):
    got = got.detach().double().cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(got - ref)
    assert err.max() <= LOWP_TOL * scale, "%s: max err %.3e vs |ref|max %.3e" % (what, err.max(), scale)
    # tighter: half an output ulp of the element (+ accumulation noise proportional to the tensor scale)
    bound = ulp_rel * np.abs(ref) + 5e-6 * scale
    assert (err <= bound).all(), "%s: exceeds rounding bound by %.3e" % (what, float((err - bound).max()))


# every kernel class: DMA ring (56-class, 28-class; aligned and 8-byte-aligned rows), register-staged (14, 7, odd sizes),
# band-skipping (5x5 on a large map), batch tails (N not a multiple of the planes per iteration), ragged channel counts
SHAPES = [
    (5, 3, 56, 56, 5, 51), (5, 3, 56, 56, 51, 5), (3, 4, 56, 56, 5, 5),
    (9, 5, 28, 28, 5, 49), (9, 5, 28, 28, 49, 5), (6, 2, 28, 28, 5, 5),
    (7, 3, 48, 40, 51, 5), (7, 3, 40, 48, 5, 51), (3, 2, 64, 64, 5, 61), (3, 2, 64, 64, 61, 5),
    (5, 4, 24, 28, 5, 31), (5, 4, 28, 24, 31, 5), (4, 3, 32, 32, 31, 5),
    (11, 5, 14, 14, 5, 47), (11, 5, 14, 14, 47, 5), (13, 3, 12, 16, 13, 5), (13, 3, 16, 12, 5, 13),
    (17, 6, 7, 7, 5, 13), (17, 6, 7, 7, 13, 5), (1, 1, 7, 7, 13, 5), (11, 2, 7, 8, 5, 9),
    (2, 130, 14, 14, 5, 5), (3, 2, 18, 20, 5, 21), (3, 2, 20, 24, 21, 5),
    # wave-independent 14x14-class kernels: single image / odd batch tails, ragged channel blocks, non-square planes, overlapping row halves
    (1, 1, 14, 14, 47, 5), (1, 1, 14, 14, 5, 47), (6, 4, 10, 12, 5, 9), (6, 4, 12, 10, 9, 5), (4, 2, 14, 8, 5, 7), (9, 7, 8, 14, 5, 5),
    (5, 3, 12, 16, 5, 13), (5, 3, 16, 12, 13, 5), (3, 9, 14, 16, 5, 31),
    # vertical weight gradient by row reads + shifted copy: first and last chunk of the tensor, non-square planes, narrow planes
    (3, 2, 56, 56, 51, 5), (1, 1, 40, 48, 31, 5), (2, 2, 64, 16, 51, 5), (2, 3, 36, 24, 35, 5),
    # vertical weight gradient, one plane per wave (planes of <= 32 rows, rows that are not whole 16-byte pieces): wave tails (N % 4),
    # slice tails, the tensor's last row, every piece count, kernels longer than the plane
    (9, 3, 28, 28, 49, 5), (1, 1, 28, 28, 49, 5), (6, 2, 32, 32, 31, 5), (13, 2, 15, 16, 13, 5), (5, 3, 20, 24, 21, 5), (130, 2, 28, 28, 49, 5),
    # wide maps (64 < long axis <= 128: banded Toeplitz, strip walk, padded-pitch plane DMA): BASELINE configs[4] stage 1 (96x96,
    # 61-tap), the 128x128 planes of the 512 px segmentation crops, non-square / non-multiple-of-32 maps, single plane
    (3, 2, 96, 96, 5, 61), (3, 2, 96, 96, 61, 5), (2, 3, 96, 96, 5, 5), (2, 2, 128, 128, 5, 61), (2, 2, 128, 128, 61, 5),
    (5, 3, 80, 96, 5, 51), (5, 3, 96, 80, 51, 5), (2, 2, 112, 72, 57, 5), (2, 2, 72, 112, 5, 57), (1, 1, 96, 96, 61, 5), (4, 2, 128, 96, 5, 31),
]


@pytest.mark.parametrize("N,C,H,W,kh,kw", SHAPES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_mfma_forward_dgrad_wgrad_vs_oracle(N, C, H, W, kh, kw, dtype, mfma_only, gpu):
    ops = _ops()
    torch.manual_seed(N * 1000 + H * 10 + kh)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dy = torch.randn(N, C, H, W, device=gpu).to(dtype)
    w = torch.randn(C, 1, kh, kw, device=gpu) * 0.05
    xr, dyr, wr = _round(x, dtype), _round(dy, dtype), _round(w, dtype)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11      # half an ulp, relative (8 / 11 significand bits)
    y = ops.dwconv2d_forward(x, w)
    assert y.dtype == dtype
    _check(y, oracle.dwconv2d_fwd(xr, wr), ulp, "fwd")
    dx = ops.dwconv2d_backward_data(dy, w)
    _check(dx, oracle.dwconv2d_bwd_data(dyr, wr), ulp, "dgrad")
    dw = ops.dwconv2d_backward_filter(dy, x, w)
    assert dw.dtype == torch.float32
    ref = oracle.dwconv2d_bwd_filter(dyr, xr, kh, kw)
    err = np.abs(dw.double().cpu().numpy() - ref).max()
    assert err <= 1e-5 * max(1.0, np.abs(ref).max()) * max(1.0, (N * H * W) ** 0.5 / 30), err    # fp32 accumulation only
    assert torch.equal(dw, ops.dwconv2d_backward_filter(dy, x, w))                                 # no atomics


@pytest.mark.parametrize("N,C,H,W,kh", [(5, 3, 20, 18, 21), (7, 2, 24, 22, 63), (3, 5, 30, 26, 7), (9, 3, 17, 30, 9), (33, 2, 28, 28, 49), (4, 1, 32, 16, 13)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("horizontal", [False, True])
def test_vertical_weight_gradient_one_plane_per_wave(N, C, H, W, kh, dtype, horizontal, gpu):
    """dwconv_mfma_wgrad_vwave_kernel (K x 5, and 5 x K with transposing reads, on planes of <= 32 rows; rows travel through registers as
    dword-aligned 16-byte pieces) on widths whose last piece holds 1, 3 or 4 dwords of the row, against the oracle; reproducible bit for bit."""
    ops = _ops()
    torch.manual_seed(N * 31 + W)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dy = torch.randn(N, C, H, W, device=gpu).to(dtype)
    kh, kw = (5, kh) if horizontal else (kh, 5)
    w = torch.randn(C, 1, kh, kw, device=gpu) * 0.05
    dw = ops.dwconv2d_backward_filter(dy, x, w)
    ref = oracle.dwconv2d_bwd_filter(_round(dy, dtype), _round(x, dtype), kh, kw)
    err = np.abs(dw.double().cpu().numpy() - ref).max()
    assert err <= 1e-5 * max(1.0, np.abs(ref).max()) * max(1.0, (N * H * W) ** 0.5 / 30), err
    assert torch.equal(dw, ops.dwconv2d_backward_filter(dy, x, w))


def test_mfma_is_what_auto_runs_for_lowp(gpu):
    """AUTO must pick the matrix-core kernels for 16-bit activations with a 5-tap side, and the fp32-exact direct
    kernels otherwise; forcing MFMA on an unsupported case must fail loudly, not fall back."""
    ops, L = _ops(), _lib()
    x = torch.randn(2, 3, 28, 28, device=gpu).bfloat16()
    w = torch.randn(3, 1, 5, 49, device=gpu) * 0.05
    L.lib().slak_set_conv_algo(L.ALGO_AUTO)
    y_auto = ops.dwconv2d_forward(x, w)
    L.lib().slak_set_conv_algo(L.ALGO_MFMA)
    try:
        y_mfma = ops.dwconv2d_forward(x, w)
        y32 = ops.dwconv2d_forward(x.float(), w)                    # fp32 activations, MFMA forced: the two-term split (tests/test_fp32_mfma_gpu.py)
        assert L.lib().slak_debug_last_kernel() == b"dwconv_mfma(f32 split)" and y32.dtype == torch.float32
        with pytest.raises(L.SlakHipError):
            ops.dwconv2d_forward(torch.randn(2, 3, 96, 96, device=gpu), w)   # fp32 on a map beyond 64: no matrix-core kernel
        with pytest.raises(L.SlakHipError):
            ops.dwconv2d_forward(x, torch.randn(3, 1, 7, 3, device=gpu))   # more rows than columns and no 5-tap side
        L.lib().slak_set_conv_algo(L.ALGO_DIRECT)
        y_direct = ops.dwconv2d_forward(x, w)
    finally:
        L.lib().slak_set_conv_algo(L.ALGO_AUTO)
    assert torch.equal(y_auto, y_mfma)
    assert not torch.equal(y_auto, y_direct)                       # direct keeps the fp32 filter: different rounding
    assert (y_auto.float() - y_direct.float()).abs().max().item() <= 1e-2 * max(1.0, y_direct.float().abs().max().item())


@pytest.mark.parametrize("N,C,H,W,kh,kw", [(2, 3, 32, 32, 3, 3), (2, 3, 32, 32, 7, 7), (3, 2, 24, 40, 13, 13), (2, 4, 56, 56, 31, 31), (2, 2, 64, 64, 27, 27),
                                           (3, 3, 14, 14, 9, 9), (4, 2, 7, 7, 3, 3), (2, 3, 28, 28, 7, 11), (2, 2, 20, 40, 9, 21), (1, 2, 64, 48, 31, 31)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_kernels_with_more_than_five_rows_run_on_the_matrix_cores(N, C, H, W, kh, kw, dtype, gpu):
    """Square kernels (the reference's own test grid: test_correctness.py:16-20, 3 .. 31) and other kernels without a 5-tap side whose rows do
    not outnumber their columns: forward and data gradient take the rows five at a time through the generic MFMA kernel (dwconv_mfma.hip,
    TALL); the weight gradient is one launch of the generic MFMA weight-gradient kernel per five filter rows.  Against the C oracle on the rounded operands."""
    ops, L = _ops(), _lib()
    torch.manual_seed(kh * 10 + H)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dy = torch.randn(N, C, H, W, device=gpu).to(dtype)
    w = torch.randn(C, 1, kh, kw, device=gpu) * (0.5 / (kh * kw) ** 0.5)
    wr = _round(w, dtype)
    y = ops.dwconv2d_forward(x, w)
    assert L.lib().slak_debug_last_kernel() == b"dwconv_mfma"
    ref = oracle.dwconv2d_fwd(_round(x, dtype), wr)
    tol = (2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11)
    assert np.abs(y.double().cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max()) + 1e-6
    dx = ops.dwconv2d_backward_data(dy, w)
    assert L.lib().slak_debug_last_kernel() == b"dwconv_mfma"
    ref = oracle.dwconv2d_bwd_data(_round(dy, dtype), wr)
    assert np.abs(dx.double().cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max()) + 1e-6
    dw = ops.dwconv2d_backward_filter(dy, x, w)                                   # one launch per five rows of the filter
    assert L.lib().slak_debug_last_kernel() == b"dwconv_mfma_wgrad" and dw.dtype == torch.float32
    ref = oracle.dwconv2d_bwd_filter(_round(dy, dtype), _round(x, dtype), kh, kw)
    assert np.abs(dw.double().cpu().numpy() - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()) * max(1.0, (N * H * W) ** 0.5 / 30)
    assert torch.equal(dw, ops.dwconv2d_backward_filter(dy, x, w))                # fixed-order reduction


@pytest.mark.parametrize("H,W,kh,kw", [(56, 56, 5, 51), (56, 56, 51, 5), (28, 28, 49, 5), (14, 14, 5, 47), (7, 7, 13, 5), (96, 96, 5, 61), (96, 96, 61, 5)])
def test_mfma_nan_stays_in_its_plane(H, W, kh, kw, mfma_only, gpu):
    """The DMA images carry no padding and planes are packed into shared MFMA tiles: a NaN/Inf in one plane must not
    reach any other plane (zero padding is applied with selects, never by multiplying)."""
    ops = _ops()
    torch.manual_seed(1)
    x = torch.randn(6, 4, H, W, device=gpu).bfloat16()
    w = torch.randn(4, 1, kh, kw, device=gpu) * 0.05
    y0 = ops.dwconv2d_forward(x, w)
    xb = x.clone()
    xb[2, 1] = float("nan")
    xb[4, 3, H - 1, W - 1] = float("inf")
    y1 = ops.dwconv2d_forward(xb, w)
    clean = torch.ones(6, 4, dtype=torch.bool, device=gpu)
    clean[2, 1] = False; clean[4, 3] = False
    assert torch.equal(y1[clean], y0[clean])
    assert torch.isnan(y1[2, 1]).all()


def test_mfma_identity_and_adjoint_full_size(mfma_only, gpu):
    ops = _ops()
    torch.manual_seed(3)
    for (N, C, H, W, kh, kw) in [(128, 96, 56, 56, 51, 5), (128, 96, 56, 56, 5, 51), (128, 192, 28, 28, 5, 49), (128, 384, 14, 14, 47, 5),
                               (64, 96, 96, 96, 61, 5), (64, 96, 96, 96, 5, 61)]:      # last two: BASELINE configs[4] stage 1 at its per-GPU batch
        x = torch.randn(N, C, H, W, device=gpu).bfloat16()
        dy = torch.randn(N, C, H, W, device=gpu).bfloat16()
        w = (torch.randn(C, 1, kh, kw, device=gpu) * 0.02).bfloat16().float()      # exactly representable filter
        wi = torch.zeros_like(w); wi[:, 0, kh // 2, kw // 2] = 1
        assert torch.equal(ops.dwconv2d_forward(x, wi), x)
        assert torch.equal(ops.dwconv2d_backward_data(x, wi), x)
        y, dx, dw = ops.dwconv2d_forward(x, w), ops.dwconv2d_backward_data(dy, w), ops.dwconv2d_backward_filter(dy, x, w)
        a = (y.double() * dy.double()).sum().item()
        b = (dx.double() * x.double()).sum().item()
        c = (dw.double() * w.double()).sum().item()
        norm = (y.double().norm() * dy.double().norm()).item()
        assert abs(a - b) <= 2e-3 * norm and abs(a - c) <= 2e-3 * norm, (a, b, c, norm)
        # a second stream must give bit-identical results (workspace is per stream)
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            y2 = ops.dwconv2d_forward(x, w)
        s.synchronize()
        assert torch.equal(y, y2)


@pytest.mark.parametrize("N,C,H,W,K", [(5, 7, 14, 14, 47), (4, 3, 12, 10, 9), (1, 1, 14, 14, 13), (3, 2, 28, 28, 49), (6, 5, 7, 7, 13),
                                       # one-launch kernels of the 56x56 / 28x28 class: batch tails, non-square planes, 8-byte-aligned rows, 64x64
                                       (5, 3, 56, 56, 51), (9, 2, 28, 28, 49), (2, 2, 48, 40, 31), (2, 3, 64, 64, 61), (3, 2, 32, 32, 31),
                                       (6, 2, 24, 24, 13), (7, 2, 28, 20, 13), (1, 1, 56, 56, 51), (11, 1, 20, 28, 49),
                                       (3, 2, 96, 96, 61)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tri_dwconv_matches_the_three_branch_convs(N, C, H, W, K, dtype, gpu):
    """block_ops.tri_dwconv (one launch for Kx5 + 5xK + 5x5 on the 14x14 class; three launches elsewhere): outputs are the
    per-branch kernels' outputs bit for bit, the input gradient is their sum (added in fp32 before the single rounding), the
    weight gradients are the per-branch ones (up to fp32 summation order where the one-launch weight-gradient kernel runs)."""
    from slak_amd import block_ops
    ops = _ops()
    torch.manual_seed(N + K)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype).requires_grad_(True)
    ws = [(torch.randn(C, 1, kh, kw, device=gpu) * 0.05).requires_grad_(True) for kh, kw in ((K, 5), (5, K), (5, 5))]
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    ys = block_ops.tri_dwconv(x, *ws)
    torch.autograd.backward(ys, dys)
    for y, w in zip(ys, ws):
        assert torch.equal(y, ops.dwconv2d_forward(x.detach(), w.detach()))
    ref_dx = sum(ops.dwconv2d_backward_data(dy, w.detach()).float() for dy, w in zip(dys, ws))
    scale = max(1.0, ref_dx.abs().max().item())
    assert (x.grad.float() - ref_dx).abs().max().item() <= 2e-2 * scale
    for dy, w in zip(dys, ws):
        # the one-launch weight gradient (slak_dwconv2d_tri_backward_filter) sums the same fp32 products in another order than the
        # per-branch kernel: equal up to fp32 summation order, bit-identical where the three per-branch launches run
        ref = ops.dwconv2d_backward_filter(dy, x.detach(), w.detach())
        assert (w.grad - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    # where the one-launch kernel ran: the summed gradient is rounded ONCE -> the oracle's half-ulp bound holds for the sum
    L = _lib()
    kind = L.lib().slak_dwconv2d_tri_supported_op(L.SLAK_BF16 if dtype == torch.bfloat16 else L.SLAK_F16, N, C, H, W, K, 1)
    if kind == 1 and (H > 32 or W > 32):                             # planes of 2 x 2 tiles: two rounded partial planes are added (tests/test_fused_launches_gpu.py)
        kind = 0
    xr = [_round(dy, dtype) for dy in dys]
    parts = [oracle.dwconv2d_bwd_data(d, _round(w.detach(), dtype)) for d, w in zip(xr, ws)]
    ref = sum(parts)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    if kind == 1:
        _check(x.grad, ref, ulp, "tri dgrad")
    else:
        assert (H, W) != (14, 14)


@pytest.mark.parametrize("N,C,H,W,K", [(5, 3, 56, 56, 51), (9, 2, 28, 28, 49), (2, 2, 48, 40, 31), (7, 2, 28, 20, 13), (3, 2, 96, 96, 61), (4, 3, 14, 14, 47)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_backward_data_accumulate_is_a_tensor_add_of_the_gradient(N, C, H, W, K, dtype, gpu):
    """slak_dwconv2d_backward_data_accumulate: dx += grad, rounded like a bf16 / fp16 tensor add -- bit-identical to computing the
    gradient into a temporary and adding it with torch (shapes whose kernel cannot accumulate take exactly that path)."""
    ops = _ops()
    torch.manual_seed(K + N)
    dy = torch.randn(N, C, H, W, device=gpu).to(dtype)
    base = torch.randn(N, C, H, W, device=gpu).to(dtype)
    for kh, kw in ((K, 5), (5, K), (5, 5)):
        w = torch.randn(C, 1, kh, kw, device=gpu) * 0.05
        want = base + ops.dwconv2d_backward_data(dy, w)
        got = ops.dwconv2d_backward_data_accumulate(dy, w, base.clone())
        assert torch.equal(got, want), (kh, kw, (got.float() - want.float()).abs().max().item())


@pytest.mark.parametrize("N,C,H,W,kh,kw", [(32, 24, 56, 56, 51, 5), (32, 24, 28, 28, 5, 49), (48, 64, 14, 14, 47, 5), (16, 8, 96, 96, 5, 61)])
def test_weight_gradient_same_shape_on_two_streams_concurrently(N, C, H, W, kh, kw, mfma_only, gpu):
    """The in-kernel slice reduction (last arriver adds the partials) takes its arrival counters from a rotating pool: launches of the
    SAME shape running concurrently on two streams must not disturb each other -- every result bit-identical to the serial one."""
    ops = _ops()
    torch.manual_seed(5)
    xs = [torch.randn(N, C, H, W, device=gpu).bfloat16() for _ in range(2)]
    dys = [torch.randn(N, C, H, W, device=gpu).bfloat16() for _ in range(2)]
    w = torch.randn(C, 1, kh, kw, device=gpu)
    want = [ops.dwconv2d_backward_filter(dys[i], xs[i], w) for i in range(2)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    got = [[], []]
    for rep in range(25):
        for i, s in enumerate(streams):
            with torch.cuda.stream(s):
                got[i].append(ops.dwconv2d_backward_filter(dys[i], xs[i], w))
    for s in streams:
        s.synchronize()
    for i in range(2):
        for g in got[i]:
            assert torch.equal(g, want[i])


def test_full_size_properties_on_slak_b_and_odd_width_channel_counts(mfma_only, gpu):
    """BASELINE configs[3] (SLaK-B: C = 128 / 256 / 512 / 1024) and the width-1.3 channel counts of the reference's
    --width_factor 1.3 recipes (124 / 249 / 499 / 998: not multiples of the 4-channel workgroups) at the per-GPU batch of the
    configuration: identity filter, adjoint identity <y, dy> = <x, dx> = <w, dw>, and 9 sampled planes against the oracle."""
    ops = _ops()
    torch.manual_seed(8)
    cases = [(64, 128, 56, 56, 51, 5), (64, 256, 28, 28, 5, 49), (64, 512, 14, 14, 47, 5), (64, 1024, 7, 7, 5, 13),
             (32, 124, 56, 56, 5, 51), (32, 249, 28, 28, 49, 5), (32, 499, 14, 14, 5, 47), (32, 998, 7, 7, 13, 5)]
    for (N, C, H, W, kh, kw) in cases:
        x = torch.randn(N, C, H, W, device=gpu).bfloat16()
        dy = torch.randn(N, C, H, W, device=gpu).bfloat16()
        w = (torch.randn(C, 1, kh, kw, device=gpu) * 0.02).bfloat16().float()
        wi = torch.zeros_like(w); wi[:, 0, kh // 2, kw // 2] = 1
        assert torch.equal(ops.dwconv2d_forward(x, wi), x)
        y, dx, dw = ops.dwconv2d_forward(x, w), ops.dwconv2d_backward_data(dy, w), ops.dwconv2d_backward_filter(dy, x, w)
        a = (y.double() * dy.double()).sum().item()
        b = (dx.double() * x.double()).sum().item()
        c = (dw.double() * w.double()).sum().item()
        norm = (y.double().norm() * dy.double().norm()).item()
        assert abs(a - b) <= 2e-3 * norm and abs(a - c) <= 2e-3 * norm, ((N, C, H, W, kh, kw), a, b, c, norm)
        g = torch.Generator().manual_seed(C)
        for _ in range(9):
            n, ch = int(torch.randint(0, N, (1,), generator=g)), int(torch.randint(0, C, (1,), generator=g))
            ref = oracle.dwconv2d_fwd(x[n:n + 1, ch:ch + 1].float().cpu().numpy(), w[ch:ch + 1].cpu().numpy())
            got = y[n, ch].double().cpu().numpy()
            assert np.abs(got - ref[0, 0]).max() <= LOWP_TOL * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("N,C,H,W,K", [(5, 7, 14, 14, 47), (4, 3, 12, 10, 9), (1, 1, 14, 14, 13), (6, 5, 7, 7, 13), (17, 6, 7, 7, 13), (1, 1, 7, 7, 13),
                                       (3, 2, 7, 5, 7), (5, 2, 5, 7, 9), (2, 130, 7, 7, 13), (9, 3, 14, 8, 31), (33, 4, 14, 14, 47), (16, 9, 6, 6, 9),
                                       # eight planes per MFMA (planes up to 7 x 7): batch tails 1..7, H < 7, W < 7, a long batch
                                       (8, 3, 7, 7, 13), (9, 2, 7, 7, 13), (15, 2, 7, 7, 13), (128, 8, 7, 7, 13), (11, 3, 5, 7, 9), (23, 2, 7, 6, 7), (131, 2, 6, 6, 13), (12, 2, 4, 4, 7)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tri_weight_gradient_matches_the_three_per_branch_kernels(N, C, H, W, K, dtype, gpu):
    """slak_dwconv2d_tri_backward_filter (one launch, x fetched once) against the oracle and the per-branch kernels: same fp32
    accumulation over the batch in a different order -> equal to accumulation noise; deterministic."""
    from slak_amd import block_ops
    ops = _ops()
    torch.manual_seed(N * 7 + K)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype).requires_grad_(True)
    ws = [(torch.randn(C, 1, kh, kw, device=gpu) * 0.05).requires_grad_(True) for kh, kw in ((K, 5), (5, K), (5, 5))]
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    got = []
    for rep in range(2):
        for w in ws:
            w.grad = None
        ys = block_ops.tri_dwconv(x, *ws)
        torch.autograd.backward(ys, dys)
        got.append([w.grad.clone() for w in ws])
    for a, b in zip(*got):
        assert torch.equal(a, b)                                                      # no atomics, fixed order
    L = _lib()
    assert L.lib().slak_dwconv2d_tri_filter_workspace_bytes(L.SLAK_BF16 if dtype == torch.bfloat16 else L.SLAK_F16, N, C, H, W, K) > 0
    for g, dy, w, (kh, kw) in zip(got[0], dys, ws, ((K, 5), (5, K), (5, 5))):
        ref = oracle.dwconv2d_bwd_filter(_round(dy, dtype), _round(x.detach(), dtype), kh, kw)
        err = np.abs(g.double().cpu().numpy() - ref).max()
        assert err <= 1e-5 * max(1.0, np.abs(ref).max()) * max(1.0, (N * H * W) ** 0.5 / 30), (kh, kw, err)
        sep = ops.dwconv2d_backward_filter(dy, x.detach(), w.detach())
        assert (g - sep).abs().max().item() <= 1e-5 * max(1.0, sep.abs().max().item()) * max(1.0, (N * H * W) ** 0.5 / 30)


@pytest.mark.parametrize("N,C,H,W,K", [(6, 5, 7, 7, 13), (40, 5, 7, 7, 13), (1, 1, 7, 7, 13), (2, 1, 7, 7, 13), (3, 4, 7, 7, 5), (33, 2, 3, 4, 7), (100, 9, 7, 7, 13),
                                       (4, 3, 6, 6, 9), (3, 2, 7, 5, 7), (7, 3, 5, 7, 31), (129, 2, 7, 7, 13), (8, 8, 7, 6, 13), (128, 16, 7, 7, 13)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_four_planes_per_tile_three_branch_kernels_on_planes_up_to_7x7(N, C, H, W, K, dtype, gpu):
    """dwconv_mfma_small_quad_kernel (dwconv_mfma_small_tri.hip: four planes of one channel per MFMA tile, block-diagonal Toeplitz
    operands) through slak_dwconv2d_tri_forward / _backward_data against the oracle: every output rounded once; batch tails of
    1..3 planes, H < 7, W = 4..7, channel tails, kernels longer than the plane, the tensor's last row (fetched early and shifted)."""
    L = _lib()
    lib = L.lib()
    dt = L.SLAK_BF16 if dtype == torch.bfloat16 else L.SLAK_F16
    torch.manual_seed(N + K + H)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    ws = [torch.randn(C, 1, kh, kw, device=gpu) * 0.05 for kh, kw in ((K, 5), (5, K), (5, 5))]
    # guard elements behind the tensors: nothing may be written there
    ybuf = [torch.full((x.numel() + 64,), 7.0, device=gpu, dtype=dtype) for _ in range(4)]
    ys = [b[:x.numel()].view_as(x) for b in ybuf[:3]]
    dx = ybuf[3][:x.numel()].view_as(x)
    st = torch.cuda.current_stream(gpu).cuda_stream
    assert lib.slak_dwconv2d_tri_supported(dt, N, C, H, W, K) == 1
    L.check(lib.slak_dwconv2d_tri_forward(x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(),
                                          ys[2].data_ptr(), dt, N, C, H, W, K, st))
    L.check(lib.slak_dwconv2d_tri_backward_data(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(),
                                                ws[2].data_ptr(), dx.data_ptr(), dt, N, C, H, W, K, st))
    torch.cuda.synchronize()
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    xr = _round(x, dtype)
    for y, w in zip(ys, ws):
        _check(y, oracle.dwconv2d_fwd(xr, _round(w, dtype)), ulp, "quad fwd")
    ref = sum(oracle.dwconv2d_bwd_data(_round(dy, dtype), _round(w, dtype)) for dy, w in zip(dys, ws))
    _check(dx, ref, ulp, "quad dgrad")
    for b in ybuf:
        assert (b[x.numel():] == 7.0).all()


@pytest.mark.parametrize("N,C,H,W,K", [(128, 32, 14, 14, 47), (6, 3, 14, 10, 13), (33, 8, 12, 14, 31),
                                       (5, 3, 56, 56, 51), (9, 2, 28, 28, 49), (2, 2, 48, 40, 31), (130, 2, 28, 28, 13),
                                       # planes up to 7 x 7: sums gathered in the store phase of the eight-planes-per-step kernel
                                       (128, 16, 7, 7, 13), (9, 5, 7, 7, 13), (11, 3, 5, 6, 9), (1, 1, 7, 7, 13), (20, 2, 4, 4, 7)])
def test_tri_forward_batch_sums_are_the_sums_of_the_stored_outputs(N, C, H, W, K, gpu):
    """slak_dwconv2d_tri_forward_stats / slak_dwconv2d_forward_stats: the partial sums the forward launches leave for the branch BatchNorms
    (models/SLaK.py:92-95) add up to sum y_b and sum y_b^2 of the three stored bf16 outputs (fp32 summation order aside), and the outputs
    are those of the plain launches."""
    from slak_amd import block_ops
    torch.manual_seed(N + K)
    x = torch.randn(N, C, H, W, device=gpu).bfloat16()
    ws = [torch.randn(C, 1, kh, kw, device=gpu) * 0.05 for kh, kw in ((K, 5), (5, K), (5, 5))]
    yv, yh, ys, st = block_ops.tri_dwconv(x, *ws, want_stats=2)
    ref = block_ops.tri_dwconv(x, *ws)
    assert all(torch.equal(a, b) for a, b in zip((yv, yh, ys), ref))
    assert len(st) == 3
    for b, y in enumerate((yv, yh, ys)):
        assert st[b].dim() == 3 and st[b].shape[0] >= 1 and st[b].shape[1] == C and st[b].shape[2] >= 2
        tot = st[b][:, :, :2].double().sum(0)
        s1 = y.double().sum((0, 2, 3)); s2 = (y.double() ** 2).sum((0, 2, 3))
        assert (tot[:, 0] - s1).abs().max().item() <= 1e-4 * max(1.0, s1.abs().max().item())
        assert (tot[:, 1] - s2).abs().max().item() <= 1e-4 * max(1.0, s2.abs().max().item())
