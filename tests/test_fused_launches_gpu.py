"""The launches the train step actually makes for a decomposed block (models/SLaK.py:82-100) -- one launch for several branches --
called THROUGH THE C ABI and compared with the C ORACLE directly (not with their sibling HIP kernels):

    slak_dwconv2d_pair_backward_filter   K x 5 and 5 x 5 weight gradient in one launch (56 x 56 / 28 x 28 class)
    slak_dwconv2d_tri_forward[_stats]    the three branch outputs in one launch
    slak_dwconv2d_tri_backward_data      the summed input gradient in one launch
    slak_dwconv2d_tri_backward_filter    the three weight gradients in one launch (14 x 14 / 7 x 7 class; round 4: planes of 2 x 2 MFMA tiles, 56 x 56 class)

Small shapes are checked exhaustively; the shapes bench.py times (BASELINE configs[1], [3], [4] at their per-GPU batch) are run at full
size and checked on a sample of >= 9 channels (first, last and seeded picks): the oracle works channel by channel, so a channel subset
of the operands is an exact restatement of those channels' results.  Operands are rounded to the activation dtype before they reach
the oracle (the kernels round the fp32 filter like autocast does): what is left is one output rounding + fp32 accumulation noise.
"""
import ctypes

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _L():
    from slak_amd import _lib
    return _lib


def _st(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _dt(dtype):
    L = _L()
    return L.SLAK_BF16 if dtype == torch.bfloat16 else L.SLAK_F16


def _r(t, dtype):
    return t.to(dtype).float().cpu().numpy()


def _filters(C, K, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return [(torch.randn(C, 1, kh, kw, generator=g) * 0.05).to(dev) for kh, kw in ((K, 5), (5, K), (5, 5))]


def _sample_channels(C, n=9, seed=0):
    if C <= n:
        return list(range(C))
    rng = np.random.default_rng(seed)
    picks = {0, C - 1} | set(int(v) for v in rng.choice(np.arange(1, C - 1), size=n - 2, replace=False))
    return sorted(picks)


def _check_lowp(got, ref, dtype, what):
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    got = got.detach().double().cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    err = np.abs(got - ref)
    assert err.max() <= 1e-2 * scale, "%s: max err %.3e (scale %.3e)" % (what, err.max(), scale)      # north star, bf16
    bound = ulp * np.abs(ref) + 5e-6 * scale                                                        # half an output ulp + accumulation
    assert (err <= bound).all(), "%s: exceeds the rounding bound by %.3e" % (what, float((err - bound).max()))


def _check_dw(got, ref, n_terms, what):
    """fp32 accumulation of n_terms products: 1e-5 of the largest gradient for short sums, growing with sqrt(n_terms), and never looser than the rtol 1e-4 the
    reference's own test allows a weight gradient (test_correctness.py:67-90) -- the stage-1 sums (401,408 terms) are held to that."""
    err = np.abs(got.double().cpu().numpy() - ref).max()
    assert err <= min(1e-4, 1e-5 * max(1.0, n_terms ** 0.5 / 30)) * max(1.0, np.abs(ref).max()), "%s: %.3e" % (what, err)


def _pair_wgrad(dyv, dys, x, K):
    L = _L(); lib = L.lib()
    N, C, H, W = x.shape
    dt = _dt(x.dtype)
    nb = int(lib.slak_dwconv2d_pair_filter_workspace_bytes(dt, N, C, H, W, K))
    if nb == 0:
        return None
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    dwv = torch.empty(C, 1, K, 5, dtype=torch.float32, device=x.device)
    dws = torch.empty(C, 1, 5, 5, dtype=torch.float32, device=x.device)
    L.check(lib.slak_dwconv2d_pair_backward_filter(dyv.data_ptr(), dys.data_ptr(), x.data_ptr(), dwv.data_ptr(), dws.data_ptr(), dt,
                                                   N, C, H, W, K, ws.data_ptr(), nb, _st(x.device)), "pair_backward_filter")
    return dwv, dws


def _tri_fwd(x, ws, K, stats=False):
    L = _L(); lib = L.lib()
    N, C, H, W = x.shape
    dt = _dt(x.dtype)
    ys = [torch.empty_like(x) for _ in range(3)]
    if stats:
        rows = int(lib.slak_dwconv2d_tri_stats_rows(dt, N, C, H, W, K))
        if rows <= 0:
            return None, None
        st = torch.zeros(rows, C, 6, dtype=torch.float32, device=x.device)
        L.check(lib.slak_dwconv2d_tri_forward_stats(x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(),
                                                    ys[1].data_ptr(), ys[2].data_ptr(), st.data_ptr(), dt, N, C, H, W, K, _st(x.device)), "tri_forward_stats")
        return ys, st
    L.check(lib.slak_dwconv2d_tri_forward(x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(),
                                          ys[1].data_ptr(), ys[2].data_ptr(), dt, N, C, H, W, K, _st(x.device)), "tri_forward")
    return ys, None


def _tri_dgrad(dys, ws, K):
    L = _L(); lib = L.lib()
    N, C, H, W = dys[0].shape
    dx = torch.empty_like(dys[0])
    L.check(lib.slak_dwconv2d_tri_backward_data(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(),
                                                ws[2].data_ptr(), dx.data_ptr(), _dt(dx.dtype), N, C, H, W, K, _st(dx.device)), "tri_backward_data")
    return dx


def _tri_wgrad(dys, x, K):
    L = _L(); lib = L.lib()
    N, C, H, W = x.shape
    dt = _dt(x.dtype)
    nb = int(lib.slak_dwconv2d_tri_filter_workspace_bytes(dt, N, C, H, W, K))
    if nb == 0:
        return None
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    dws = [torch.empty(C, 1, kh, kw, dtype=torch.float32, device=x.device) for kh, kw in ((K, 5), (5, K), (5, 5))]
    L.check(lib.slak_dwconv2d_tri_backward_filter(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(),
                                                  dws[1].data_ptr(), dws[2].data_ptr(), dt, N, C, H, W, K, ws.data_ptr(), nb, _st(x.device)), "tri_backward_filter")
    return dws


def _tri_bwd(dys, x, ws, K):
    """dx and the three weight gradients in ONE launch (slak_dwconv2d_tri_backward); None where there is no such launch"""
    L = _L(); lib = L.lib()
    N, C, H, W = x.shape
    dt = _dt(x.dtype)
    if not lib.slak_dwconv2d_tri_backward_supported(dt, N, C, H, W, K):
        return None
    nb = int(lib.slak_dwconv2d_tri_filter_workspace_bytes(dt, N, C, H, W, K))
    assert nb > 0
    wsb = torch.empty(nb, dtype=torch.uint8, device=x.device)
    dx = torch.full_like(x, float("nan"))
    dws = [torch.full((C, 1, kh, kw), float("nan"), dtype=torch.float32, device=x.device) for kh, kw in ((K, 5), (5, K), (5, 5))]
    L.check(lib.slak_dwconv2d_tri_backward(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(),
                                           ws[2].data_ptr(), dx.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), dt, N, C, H, W, K,
                                           wsb.data_ptr(), nb, _st(x.device)), "tri_backward")
    return dx, dws


# ------------------------------------------------------------------------------------------------ small shapes, every channel
PAIR_SMALL = [(5, 3, 56, 56, 51), (3, 2, 56, 56, 51), (1, 1, 56, 56, 51), (9, 3, 28, 28, 49), (1, 1, 28, 28, 49), (11, 2, 28, 28, 49),
              (2, 2, 64, 16, 51), (2, 3, 36, 24, 35), (1, 1, 40, 48, 31), (6, 2, 32, 32, 31), (5, 3, 20, 24, 21), (13, 2, 15, 16, 13),
              (4, 3, 24, 24, 57), (3, 2, 48, 48, 59), (3, 1, 64, 64, 61)]


@pytest.mark.parametrize("N,C,H,W,K", PAIR_SMALL)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_pair_backward_filter_vs_oracle(N, C, H, W, K, dtype, gpu):
    torch.manual_seed(N * 100 + H + K)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dyv, dys = (torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(2))
    r = _pair_wgrad(dyv, dys, x, K)
    assert r is not None, "every shape of PAIR_SMALL has a two-branch weight-gradient launch"
    dwv, dws = r
    xr = _r(x, dtype)
    _check_dw(dwv, oracle.dwconv2d_bwd_filter(_r(dyv, dtype), xr, K, 5), N * H * W, "dw Kx5")
    _check_dw(dws, oracle.dwconv2d_bwd_filter(_r(dys, dtype), xr, 5, 5), N * H * W, "dw 5x5")
    again = _pair_wgrad(dyv, dys, x, K)
    assert torch.equal(dwv, again[0]) and torch.equal(dws, again[1])        # fixed-order slice reduction, no atomics on the data


TRI_SMALL = [(5, 7, 14, 14, 47), (4, 3, 12, 10, 9), (1, 1, 14, 14, 13), (6, 5, 7, 7, 13), (17, 6, 7, 7, 13), (3, 9, 14, 14, 47), (9, 4, 12, 12, 13),
             (5, 3, 56, 56, 51), (1, 1, 56, 56, 51), (7, 2, 56, 56, 51), (9, 2, 28, 28, 49), (1, 1, 28, 28, 49), (11, 3, 28, 28, 49),
             (2, 2, 48, 40, 31), (3, 2, 32, 32, 31), (6, 2, 24, 24, 13), (5, 2, 24, 24, 57), (7, 2, 28, 20, 13), (3, 2, 48, 48, 59)]


def test_shapes_without_a_three_branch_launch_say_so(gpu):
    """Maps beyond 96 (round 6: 64 < H, W <= 96 in multiples of 16 have the wide three-branch kernel), planes whose width is not a multiple of 8 above 32, fp32: the support query is 0 and the entry points refuse
    instead of running something else (block_ops.tri_dwconv then issues the three per-branch launches:
    tests/test_mfma_gpu.py::test_tri_dwconv_matches_the_three_branch_convs covers those shapes).  64 x 64 planes have a forward
    kernel (the stream kernel) and no data-gradient one: each op answers for itself."""
    L = _L(); lib = L.lib()
    for (N, C, H, W, K) in [(3, 2, 128, 128, 61), (2, 2, 36, 36, 31), (2, 2, 96, 88, 61)]:
        assert lib.slak_dwconv2d_tri_supported(L.SLAK_BF16, N, C, H, W, K) == 0
        x = torch.randn(N, C, H, W, device=gpu).bfloat16()
        ws = _filters(C, K, gpu, 1)
        ys = [torch.empty_like(x) for _ in range(3)]
        rc = lib.slak_dwconv2d_tri_forward(x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(),
                                           ys[2].data_ptr(), L.SLAK_BF16, N, C, H, W, K, _st(gpu))
        assert rc == L.ERR_UNSUPPORTED, (N, C, H, W, K, rc)
    N, C, H, W, K = 2, 3, 64, 64, 61
    assert lib.slak_dwconv2d_tri_supported(L.SLAK_BF16, N, C, H, W, K) == 0
    assert lib.slak_dwconv2d_tri_supported_op(L.SLAK_BF16, N, C, H, W, K, 1) == 0
    dys = [torch.randn(N, C, H, W, device=gpu).bfloat16() for _ in range(3)]
    ws = _filters(C, K, gpu, 1)
    dx = torch.empty_like(dys[0])
    rc = lib.slak_dwconv2d_tri_backward_data(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(),
                                             ws[2].data_ptr(), dx.data_ptr(), L.SLAK_BF16, N, C, H, W, K, _st(gpu))
    assert rc == L.ERR_UNSUPPORTED, rc
    if lib.slak_dwconv2d_tri_supported_op(L.SLAK_BF16, N, C, H, W, K, 0) == 1:
        x = torch.randn(N, C, H, W, device=gpu).bfloat16()
        ys, _ = _tri_fwd(x, ws, K)
        xr = _r(x, torch.bfloat16)
        for y, w, name in zip(ys, ws, ("Kx5", "5xK", "5x5")):
            _check_lowp(y, oracle.dwconv2d_fwd(xr, _r(w, torch.bfloat16)), torch.bfloat16, "tri fwd 64x64 " + name)
    assert lib.slak_dwconv2d_tri_supported_op(L.SLAK_F32, 4, 4, 14, 14, 47, 0) == 0


@pytest.mark.parametrize("N,C,H,W,K", TRI_SMALL)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tri_forward_and_backward_data_vs_oracle(N, C, H, W, K, dtype, gpu):
    L = _L()
    assert L.lib().slak_dwconv2d_tri_supported(_dt(dtype), N, C, H, W, K) == 1
    torch.manual_seed(N * 100 + H + K)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    ws = _filters(C, K, gpu, K + C)
    wr = [_r(w, dtype) for w in ws]
    ys, _ = _tri_fwd(x, ws, K)
    xr = _r(x, dtype)
    for y, w, name in zip(ys, wr, ("Kx5", "5xK", "5x5")):
        _check_lowp(y, oracle.dwconv2d_fwd(xr, w), dtype, "tri fwd " + name)
    dx = _tri_dgrad(dys, ws, K)
    parts = [oracle.dwconv2d_bwd_data(_r(d, dtype), w) for d, w in zip(dys, wr)]
    ref = sum(parts)
    got = dx.detach().double().cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert np.abs(got - ref).max() <= 1e-2 * scale
    # a kernel may round each branch's partial once before the fp32 sum is rounded (what autograd's tensor adds do); one that adds in its
    # accumulator rounds only the sum: half an ulp of every value that is rounded, plus accumulation noise
    bound = ulp * (np.abs(ref) + sum(np.abs(p) for p in parts)) + 5e-6 * scale
    assert (np.abs(got - ref) <= bound).all(), float((np.abs(got - ref) - bound).max())


# round 6: maps with 64 < H, W <= 96 (dwconv_mfma_wide_tri.hip): the 96 x 96 planes of SLaK at 384 px, 80-wide maps (two and a half tiles: lanes beyond
# the map), short filters (narrow band: fewer blocks per tile), and a batch that makes workgroup ranges cross a channel boundary (fragments rebuilt in mid-stream)
TRI_WIDE = [(3, 2, 96, 96, 61), (1, 1, 96, 96, 13), (2, 2, 80, 96, 33), (2, 3, 96, 80, 51), (2, 2, 80, 80, 61), (131, 3, 80, 80, 31)]


@pytest.mark.parametrize("N,C,H,W,K", TRI_WIDE)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tri_forward_and_backward_data_on_wide_maps_vs_oracle(N, C, H, W, K, dtype, gpu):
    L = _L()
    assert L.lib().slak_dwconv2d_tri_supported(_dt(dtype), N, C, H, W, K) == 1
    torch.manual_seed(N * 100 + H + K)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    ws = _filters(C, K, gpu, K + C)
    wr = [_r(w, dtype) for w in ws]
    ys, _ = _tri_fwd(x, ws, K)
    assert L.lib().slak_debug_last_kernel().decode() == "dwconv_mfma_wide_tri"
    xr = _r(x, dtype)
    for y, w, name in zip(ys, wr, ("Kx5", "5xK", "5x5")):
        _check_lowp(y, oracle.dwconv2d_fwd(xr, w), dtype, "wide tri fwd " + name)
    dx = _tri_dgrad(dys, ws, K)
    assert L.lib().slak_debug_last_kernel().decode() == "dwconv_mfma_wide_tri"
    parts = [oracle.dwconv2d_bwd_data(_r(d, dtype), w) for d, w in zip(dys, wr)]
    ref = sum(parts)
    got = dx.detach().double().cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    assert np.abs(got - ref).max() <= 1e-2 * scale
    bound = ulp * np.abs(ref) + 1e-5 * scale                        # the three partial gradients are added in the accumulator: ONE rounding
    assert (np.abs(got - ref) <= bound).all(), float((np.abs(got - ref) - bound).max())
    # run to run identical, and a non-finite value stays inside its own plane (the slots' guard rows and the lanes beyond the map never reach another plane's sums)
    ys2, _ = _tri_fwd(x, ws, K)
    assert all(torch.equal(a, b) for a, b in zip(ys, ys2)) and torch.equal(dx, _tri_dgrad(dys, ws, K))
    if N > 1:
        xn = x.clone(); xn[0, 0, H - 1, W - 1] = float("nan"); xn[0, 0, 0, 0] = float("inf")
        yn, _ = _tri_fwd(xn, ws, K)
        for a, b in zip(ys, yn):
            assert torch.equal(a[1:], b[1:]) and (C == 1 or torch.equal(a[0, 1:], b[0, 1:]))


@pytest.mark.parametrize("N,C,H,W,K", [(5, 7, 14, 14, 47), (6, 5, 7, 7, 13), (17, 6, 7, 7, 13), (4, 3, 12, 10, 9), (9, 4, 12, 12, 13), (1, 1, 14, 14, 13)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tri_backward_filter_vs_oracle(N, C, H, W, K, dtype, gpu):
    torch.manual_seed(N + K)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    dws = _tri_wgrad(dys, x, K)
    assert dws is not None
    xr = _r(x, dtype)
    for dw, dy, (kh, kw) in zip(dws, dys, ((K, 5), (5, K), (5, 5))):
        _check_dw(dw, oracle.dwconv2d_bwd_filter(_r(dy, dtype), xr, kh, kw), N * H * W, "dw %dx%d" % (kh, kw))


# the block's whole conv backward in one launch (14 x 14 class): the SAME bits as the data-gradient launch and the weight-gradient launch (which
# the tests above hold against the oracle), on odd and even slices, one image, one channel, every supported width, more channels than a workgroup's four
TRI_BWD = [(5, 7, 14, 14, 47), (9, 4, 12, 12, 13), (4, 3, 12, 10, 9), (1, 1, 14, 14, 13), (2, 5, 14, 8, 27), (33, 6, 9, 14, 31), (128, 9, 14, 14, 47),
           (7, 2, 8, 8, 7), (16, 4, 14, 12, 63)]


@pytest.mark.parametrize("N,C,H,W,K", TRI_BWD)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tri_backward_in_one_launch_is_the_two_launches(N, C, H, W, K, dtype, gpu):
    torch.manual_seed(3 * N + K)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    ws = _filters(C, K, gpu, K + N)
    got = _tri_bwd(dys, x, ws, K)
    assert got is not None, "no one-launch backward for a 14 x 14 class shape"
    dx, dws = got
    assert torch.equal(dx, _tri_dgrad(dys, ws, K))
    for a, b in zip(dws, _tri_wgrad(dys, x, K)):
        assert torch.equal(a, b)
    # and against the oracle directly (the data gradient with the two-rounding bound of the tests above)
    wr = [_r(w, dtype) for w in ws]
    parts = [oracle.dwconv2d_bwd_data(_r(d, dtype), w) for d, w in zip(dys, wr)]
    ref = sum(parts)
    ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    bound = ulp * (np.abs(ref) + sum(np.abs(p) for p in parts)) + 5e-6 * max(1.0, float(np.abs(ref).max()))
    assert (np.abs(dx.double().cpu().numpy() - ref) <= bound).all()
    xr = _r(x, dtype)
    for dw, dy, (kh, kw) in zip(dws, dys, ((K, 5), (5, K), (5, 5))):
        _check_dw(dw, oracle.dwconv2d_bwd_filter(_r(dy, dtype), xr, kh, kw), N * H * W, "one-launch dw %dx%d" % (kh, kw))


def test_tri_backward_in_one_launch_says_where_it_does_not_exist(gpu):
    L = _L(); lib = L.lib()
    for (N, C, H, W, K) in [(4, 4, 7, 7, 13), (4, 4, 28, 28, 49), (4, 4, 56, 56, 51), (4, 4, 14, 7, 13), (4, 4, 14, 13, 13)]:
        assert lib.slak_dwconv2d_tri_backward_supported(L.SLAK_BF16, N, C, H, W, K) == 0, (N, C, H, W, K)
        x = torch.randn(N, C, H, W, device=gpu).bfloat16()
        ws = _filters(C, K, gpu, 1)
        dws = [torch.empty(C, 1, kh, kw, dtype=torch.float32, device=gpu) for kh, kw in ((K, 5), (5, K), (5, 5))]
        wsb = torch.empty(1 << 20, dtype=torch.uint8, device=gpu)
        dx = torch.empty_like(x)
        rc = lib.slak_dwconv2d_tri_backward(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(),
                                            dx.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), L.SLAK_BF16, N, C, H, W, K,
                                            wsb.data_ptr(), wsb.numel(), _st(gpu))
        assert rc == L.ERR_UNSUPPORTED, (N, C, H, W, K, rc)
    assert lib.slak_dwconv2d_tri_backward_supported(L.SLAK_F32, 4, 4, 14, 14, 47) == 0


# planes of 2 x 2 MFMA tiles (33 .. 64, W % 8 == 0): dwconv_mfma_tri_wgrad_rows.hip.  One workgroup per CU owns a RANGE of the C * N planes in
# (channel, image) order; (129, 3), (67, 5) and (9, 37) make ranges that cross channel boundaries with 256 workgroups (the accumulators are summed
# up and cleared in mid-stream, the first channel's sums leave with the last record), (300, 1) gives workgroups of several planes of one channel
TRI_ROWS = [(5, 3, 56, 56, 51), (1, 1, 56, 56, 51), (7, 2, 56, 56, 51), (2, 2, 64, 64, 61), (3, 2, 48, 48, 59), (2, 3, 40, 48, 31), (2, 2, 56, 40, 13),
            (7, 2, 64, 56, 51), (129, 3, 56, 56, 51), (67, 5, 56, 56, 51), (9, 37, 48, 48, 59), (300, 1, 56, 56, 51),
            # round 6, the 5 x 5 branch on the diagonal tiles (KS = 4, H <= 62): the second X tile starts at row H - 32 -- overlaps of 24, 30 and 2 rows, 64 columns
            (3, 2, 40, 56, 27), (2, 2, 34, 64, 31), (2, 2, 62, 56, 51), (5, 2, 50, 56, 9)]


@pytest.mark.parametrize("N,C,H,W,K", TRI_ROWS)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tri_backward_filter_on_planes_of_2x2_tiles_vs_oracle(N, C, H, W, K, dtype, gpu):
    torch.manual_seed(N + K + W)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    dws = _tri_wgrad(dys, x, K)
    assert dws is not None, "every shape of TRI_ROWS has a one-launch weight gradient"
    assert _L().lib().slak_debug_last_kernel() == b"dwconv_mfma_tri_wgrad_rows"
    xr = _r(x, dtype)
    for dw, dy, (kh, kw) in zip(dws, dys, ((K, 5), (5, K), (5, 5))):
        _check_dw(dw, oracle.dwconv2d_bwd_filter(_r(dy, dtype), xr, kh, kw), N * H * W, "dw %dx%d" % (kh, kw))
    again = _tri_wgrad(dys, x, K)
    assert all(torch.equal(a, b) for a, b in zip(dws, again))              # fixed-order reduction over the workgroups of a channel
    # the one-launch gradients against the per-branch launches (other kernels, other summation order): fp32 accumulation noise only
    from slak_amd import ops
    for dw, dy, (kh, kw) in zip(dws, dys, ((K, 5), (5, K), (5, 5))):
        ref = ops.dwconv2d_backward_filter(dy, x, torch.empty(C, 1, kh, kw, device=gpu))
        assert (dw - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item()) * max(1.0, (N * H * W) ** 0.5 / 30)


# planes of one MFMA tile (15 .. 32 rows, W even 16 .. 32): dwconv_mfma_tri_wgrad_wave.hip.  Wave-granular, channel-aligned work split: (130, 7) and
# (2, 300) give channels an uneven number of waves resp. more channels than ... waves per channel = 1
TRI_WAVE = [(9, 2, 28, 28, 49), (1, 1, 28, 28, 49), (11, 3, 28, 28, 49), (5, 2, 24, 24, 57), (7, 2, 28, 20, 13), (3, 2, 32, 32, 31), (4, 3, 16, 32, 31),
            (6, 2, 20, 24, 21), (130, 7, 28, 28, 49), (2, 300, 24, 24, 13), (1, 1030, 28, 28, 13)]


@pytest.mark.parametrize("N,C,H,W,K", TRI_WAVE)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tri_backward_filter_on_one_tile_planes_vs_oracle(N, C, H, W, K, dtype, gpu):
    torch.manual_seed(N + K + W)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    dws = _tri_wgrad(dys, x, K)
    if C > 1024:                                                     # more channels than waves (4 x 256 CUs): no one-launch kernel, the support query says so
        assert dws is None
        return
    assert dws is not None, "every shape of TRI_WAVE has a one-launch weight gradient"
    assert _L().lib().slak_debug_last_kernel() == b"dwconv_mfma_tri_wgrad_wave"
    xr = _r(x, dtype)
    for dw, dy, (kh, kw) in zip(dws, dys, ((K, 5), (5, K), (5, 5))):
        _check_dw(dw, oracle.dwconv2d_bwd_filter(_r(dy, dtype), xr, kh, kw), N * H * W, "dw %dx%d" % (kh, kw))
    again = _tri_wgrad(dys, x, K)
    assert all(torch.equal(a, b) for a, b in zip(dws, again))              # fixed-order reduction over the waves of a channel


def test_tri_backward_filter_keeps_padding_out_of_the_sums(gpu):
    """Rows / columns behind the image are zero PADDING of the LDS images: a non-finite value in the LAST row and column of a plane must only
    reach the taps that touch it -- the gradient of every other plane and channel, and the taps of this one that never see the corner, stay finite."""
    N, C, H, W, K = 3, 2, 56, 56, 51
    torch.manual_seed(0)
    x = torch.randn(N, C, H, W, device=gpu).bfloat16()
    dys = [torch.randn(N, C, H, W, device=gpu).bfloat16() for _ in range(3)]
    x[1, 0, H - 1, W - 1] = float("inf")
    dws = _tri_wgrad(dys, x, K)
    assert all(torch.isfinite(dw[1]).all() for dw in dws)                  # the other channel
    assert not all(torch.isfinite(dw[0]).all() for dw in dws)              # the channel that holds it


# ------------------------------------------------------------------------------------------------ the shapes bench.py times, sampled channels
BENCH_PAIR = [(128, 96, 56, 56, 51), (128, 192, 28, 28, 49),             # BASELINE configs[1] / [2] stages 1, 2
              (64, 128, 56, 56, 51), (64, 256, 28, 28, 49),              # configs[3] SLaK-B
              (64, 192, 48, 48, 59), (64, 384, 24, 24, 57)]              # configs[4] 61 x 61 at 384 px, stages 2, 3
BENCH_TRI = [(128, 96, 56, 56, 51), (128, 192, 28, 28, 49), (128, 384, 14, 14, 47), (128, 768, 7, 7, 13),
             (64, 128, 56, 56, 51), (64, 512, 14, 14, 47), (64, 1024, 7, 7, 13),
             (64, 192, 48, 48, 59), (64, 384, 24, 24, 57), (64, 768, 12, 12, 13),
             (64, 96, 96, 96, 61)]                                  # round 6: configs[4] stage 1 (dwconv_mfma_wide_tri: forward and data gradient in one launch each)


@pytest.mark.parametrize("N,C,H,W,K", BENCH_PAIR)
def test_pair_backward_filter_at_bench_shapes(N, C, H, W, K, gpu):
    dtype = torch.bfloat16
    torch.manual_seed(K)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dyv, dys = (torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(2))
    r = _pair_wgrad(dyv, dys, x, K)
    assert r is not None
    dwv, dws = r
    ch = _sample_channels(C, 9, seed=K)
    xr = _r(x[:, ch], dtype)
    _check_dw(dwv[ch], oracle.dwconv2d_bwd_filter(_r(dyv[:, ch], dtype), xr, K, 5), N * H * W, "dw Kx5")
    _check_dw(dws[ch], oracle.dwconv2d_bwd_filter(_r(dys[:, ch], dtype), xr, 5, 5), N * H * W, "dw 5x5")
    # size-independent property on ALL channels: with dy = x the centre tap of both gradients is sum x^2 per channel
    dwv2, dws2 = _pair_wgrad(x, x, x, K)
    e = (x.double() ** 2).sum(dim=(0, 2, 3))
    assert (dwv2[:, 0, K // 2, 2].double() - e).abs().max().item() <= 1e-4 * e.max().item()
    assert (dws2[:, 0, 2, 2].double() - e).abs().max().item() <= 1e-4 * e.max().item()


@pytest.mark.parametrize("N,C,H,W,K", BENCH_TRI)
def test_tri_launches_at_bench_shapes(N, C, H, W, K, gpu):
    dtype = torch.bfloat16
    L = _L()
    assert L.lib().slak_dwconv2d_tri_supported(_dt(dtype), N, C, H, W, K) == 1
    torch.manual_seed(K + 1)
    x = torch.randn(N, C, H, W, device=gpu).to(dtype)
    dys = [torch.randn(N, C, H, W, device=gpu).to(dtype) for _ in range(3)]
    ws = _filters(C, K, gpu, K)
    ch = _sample_channels(C, 9, seed=K)
    wr = [_r(w[ch], dtype) for w in ws]
    xr = _r(x[:, ch], dtype)
    ys, _ = _tri_fwd(x, ws, K)
    for y, w, name in zip(ys, wr, ("Kx5", "5xK", "5x5")):
        _check_lowp(y[:, ch], oracle.dwconv2d_fwd(xr, w), dtype, "tri fwd " + name)
    ys2, st = _tri_fwd(x, ws, K, stats=True)
    if ys2 is not None:                                                   # the statistics variant stores the same outputs
        for a, b in zip(ys, ys2):
            assert torch.equal(a, b)
        s = st.double().sum(dim=0)                                       # [C][6]
        for b, y in enumerate(ys2):
            yd = y.double()
            s1, s2 = yd.sum(dim=(0, 2, 3)), (yd * yd).sum(dim=(0, 2, 3))
            assert (s[:, 2 * b] - s1).abs().max().item() <= 1e-4 * max(1.0, s2.max().item() ** 0.5 * (N * H * W) ** 0.5)
            assert (s[:, 2 * b + 1] - s2).abs().max().item() <= 1e-4 * s2.max().item()
    dx = _tri_dgrad(dys, ws, K)
    parts = [oracle.dwconv2d_bwd_data(_r(d[:, ch], dtype), w) for d, w in zip(dys, wr)]
    ref = sum(parts)
    got = dx[:, ch].double().cpu().numpy()
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(got - ref).max() <= 1e-2 * scale
    # the same two-rounding bound as on the small shapes (test_tri_forward_and_backward_data_vs_oracle): half an ulp of every value a kernel
    # may round (each branch's partial, then the sum) + fp32 accumulation noise
    bound = 2.0 ** -8 * (np.abs(ref) + sum(np.abs(p) for p in parts)) + 5e-6 * scale
    assert (np.abs(got - ref) <= bound).all(), float((np.abs(got - ref) - bound).max())
    # identity filters on ALL channels: forward returns x three times, the data gradient the sum of its inputs rounded like tensor adds
    wi = [torch.zeros_like(w) for w in ws]
    wi[0][:, 0, K // 2, 2] = 1; wi[1][:, 0, 2, K // 2] = 1; wi[2][:, 0, 2, 2] = 1
    yi, _ = _tri_fwd(x, wi, K)
    assert all(torch.equal(y, x) for y in yi)
    dxi = _tri_dgrad(dys, wi, K).float()
    want = dys[0].float() + dys[1].float() + dys[2].float()
    assert (dxi - want).abs().max().item() <= 2.0 ** -7 * max(1.0, want.abs().max().item())
    dws = _tri_wgrad(dys, x, K)
    if dws is not None:
        for dw, dy, (kh, kw) in zip(dws, dys, ((K, 5), (5, K), (5, 5))):
            _check_dw(dw[ch], oracle.dwconv2d_bwd_filter(_r(dy[:, ch], dtype), xr, kh, kw), N * H * W, "dw %dx%d" % (kh, kw))
        # size-independent property on ALL channels of the one-launch weight gradient (the work split of the rows / wave kernels depends on C * N and the CU
        # count: every channel's record has to come out whichever workgroups shared it): with dy = x on the three branches the centre tap of each of the three
        # gradients is sum x^2 of the channel
        dwi = _tri_wgrad([x, x, x], x, K)
        e = (x.double() ** 2).sum(dim=(0, 2, 3))
        for dw, (ci, cj), name in zip(dwi, ((K // 2, 2), (2, K // 2), (2, 2)), ("Kx5", "5xK", "5x5")):
            assert (dw[:, 0, ci, cj].double() - e).abs().max().item() <= 1e-4 * e.max().item(), name


def test_tri_backward_in_one_launch_on_random_shapes(gpu):
    """Forty seeded random shapes of the 14 x 14 class (1..40 images, 1..13 channels, 4..14 rows, even widths 8..14, odd K 7..63): the one-launch backward
    reproduces the data-gradient launch and the weight-gradient launch bit for bit."""
    import random
    rnd = random.Random(20260927)
    for _ in range(40):
        N, C, H, W, K = rnd.randint(1, 40), rnd.randint(1, 13), rnd.randint(4, 14), 2 * rnd.randint(4, 7), 2 * rnd.randint(3, 31) + 1
        torch.manual_seed(N * 1000 + C * 100 + H + W + K)
        x = torch.randn(N, C, H, W, device=gpu).bfloat16()
        dys = [torch.randn(N, C, H, W, device=gpu).bfloat16() for _ in range(3)]
        ws = _filters(C, K, gpu, K + N)
        got = _tri_bwd(dys, x, ws, K)
        assert got is not None, (N, C, H, W, K)
        dx, dws = got
        assert torch.equal(dx, _tri_dgrad(dys, ws, K)), (N, C, H, W, K)
        for a, b in zip(dws, _tri_wgrad(dys, x, K)):
            assert torch.equal(a, b), (N, C, H, W, K)
