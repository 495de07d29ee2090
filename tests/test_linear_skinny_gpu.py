"""slak_linear_nt (csrc/linear_skinny.hip): the pointwise convolutions of the large maps as streaming kernels.  Checker: the same
product evaluated in fp64 on the bf16 operands (torch CPU), the result rounded once -- half an ulp of bf16 plus fp32 accumulation
noise; GELU = nn.GELU() (exact erf) of the ROUNDED pre-activation, as F.gelu of a bf16 tensor computes it."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,K,gelu", [(401, 384, 96, True), (3136, 384, 96, True), (100000, 384, 96, True), (257, 96, 384, False),
                                        (70001, 96, 384, False), (97, 384, 96, False), (33, 96, 192, False), (64, 96, 96, False),
                                        (31, 192, 96, True), (129, 64, 96, False), (95, 128, 96, True), (5000, 96, 288, False)])
@pytest.mark.parametrize("with_bias", [True, False])
def test_linear_nt_matches_fp64_product(M, N, K, gelu, with_bias, gpu):
    from slak_amd import block_ops
    block_ops.use_skinny_linear = True
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device=gpu).bfloat16()
    wt = (torch.randn(N, K, device=gpu) * 0.1).bfloat16()
    bias = torch.randn(N, device=gpu).bfloat16() if with_bias else None
    r = block_ops.linear_nt(x, wt, bias, gelu=gelu)
    assert r is not None, "shape must be covered"
    y = r[0] if gelu else r
    ref = x.double().cpu() @ wt.double().cpu().t()
    if with_bias:
        ref = ref + bias.double().cpu()
    err = (y.double().cpu() - ref).abs()
    bound = 2.0 ** -8 * ref.abs() + 1e-5 * max(1.0, ref.abs().max().item())
    assert (err <= bound).all(), float((err - bound).max())
    if gelu:
        want = torch.nn.functional.gelu(y.float()).to(torch.bfloat16)          # gelu of the rounded pre-activation, rounded once
        d = (r[1].float() - want.float()).abs()
        assert (d <= 2.0 ** -7 * want.float().abs() + 1e-6).all(), d.max().item()      # at most one bf16 ulp (erf implementations differ in the last bit)
        assert (d == 0).float().mean().item() > 0.98


def test_linear_nt_declines_what_it_does_not_cover(gpu):
    from slak_amd import block_ops
    block_ops.use_skinny_linear = True
    x = torch.randn(64, 384, device=gpu).bfloat16()
    assert block_ops.linear_nt(x, torch.randn(1536, 384, device=gpu).bfloat16()) is None        # stage-3 shape: library GEMM
    assert block_ops.linear_nt(torch.randn(64, 192, device=gpu).bfloat16(), torch.randn(768, 192, device=gpu).bfloat16()) is None   # stage 2: library GEMM
    assert block_ops.linear_nt(x.float(), torch.randn(96, 384, device=gpu)) is None


def test_mlp_with_skinny_linears_matches_library_path(gpu):
    """The block MLP through the streaming kernels vs the same autograd node through the library GEMMs: forward and every gradient
    within bf16 rounding of each other."""
    from slak_amd import block_ops
    torch.manual_seed(3)
    C, M = 96, 2 * 56 * 56
    t = torch.randn(M, C, device=gpu).bfloat16()
    w1 = (torch.randn(4 * C, C, device=gpu) * 0.05).requires_grad_(True); b1 = torch.randn(4 * C, device=gpu).requires_grad_(True)
    w2 = (torch.randn(C, 4 * C, device=gpu) * 0.05).requires_grad_(True); b2 = torch.randn(C, device=gpu).requires_grad_(True)
    dz = torch.randn(M, C, device=gpu).bfloat16()
    res = {}
    for mode in (True, False):
        block_ops.use_skinny_linear = mode
        try:
            ti = t.clone().requires_grad_(True)
            for p in (w1, b1, w2, b2):
                p.grad = None
            z = block_ops.mlp_splitk(ti, w1, b1, w2, b2)
            z.backward(dz)
            res[mode] = [z.detach().float(), ti.grad.float(), w1.grad.clone(), b1.grad.clone(), w2.grad.clone(), b2.grad.clone()]
        finally:
            block_ops.use_skinny_linear = True
    for a, b, n in zip(res[True], res[False], ("z", "dt", "dw1", "db1", "dw2", "db2")):
        scale = max(1.0, b.abs().max().item())
        assert (a - b).abs().max().item() <= 2e-2 * scale, n


WGRAD_SHAPES = [(6272, 384, 96), (6272, 96, 384), (3136, 768, 192), (3136, 192, 768), (1568, 1536, 384), (1568, 384, 1536), (784, 3072, 768),
                (1000, 384, 96), (777, 192, 192), (32, 192, 384), (128 * 196, 1536, 384),
                # round 6: SLaK-B's widths (128 * 2^k) on the 128 x 64 wave tiles
                (6272, 512, 128), (6272, 128, 512), (3136, 1024, 256), (1568, 512, 2048), (784, 4096, 1024), (1000, 256, 1024), (64 * 196, 2048, 512), (40, 128, 256)]


@pytest.mark.parametrize("M,N1,N2", WGRAD_SHAPES)
def test_linear_wgrad_matches_fp32(M, N1, N2, gpu):
    """slak_linear_wgrad = dY^T X in fp32 (models/SLaK.py:117-118 weight gradients): bf16 products are exact in fp32, so the only
    difference from the fp64 reference is fp32 summation order: 1e-5 of the result's scale."""
    from slak_amd import block_ops
    torch.manual_seed(M + N1)
    dy = torch.randn(M, N1, device=gpu).bfloat16()
    x = (torch.randn(M, N2, device=gpu) + 0.25).bfloat16()
    d = block_ops.linear_wgrad(dy, x)
    assert d is not None and d.dtype == torch.float32 and d.shape == (N1, N2)
    ref = dy.double().t() @ x.double()
    err = (d.double() - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item() + 1e-6, err
    d2 = block_ops.linear_wgrad(dy, x)
    assert torch.equal(d, d2), "fixed summation order: bitwise repeatable"


def test_linear_wgrad_unsupported_shapes_fall_back(gpu):
    from slak_amd import block_ops, _lib
    L = _lib.lib()
    assert not L.slak_linear_wgrad_supported(4096, 160, 512) and not L.slak_linear_wgrad_supported(16, 384, 96)
    assert block_ops.linear_wgrad(torch.zeros(4096, 160, device=gpu).bfloat16(), torch.zeros(4096, 512, device=gpu).bfloat16()) is None




@pytest.mark.parametrize("M", [32, 96, 6272, 40032, 401408])
def test_linear_nt_gelu_bwd_is_the_gemm_followed_by_the_gelu_backward(M, gpu):
    """slak_linear_nt_gelu_bwd (stage 1: K = 96, N = 384): dy1 bit for bit what slak_linear_nt + slak_gelu_backward_bias store -- zeros, tiny,
    huge, infinite and NaN pre-activations included (the general evaluation) -- and the bias gradient = column sums of the stored dy1."""
    from slak_amd import block_ops, _lib
    L = _lib.lib()
    N, K = 384, 96
    assert L.slak_linear_nt_gelu_bwd_supported(M, N, K) == 1
    torch.manual_seed(M)
    dz = (torch.randn(M, K, device=gpu) * 0.5).bfloat16()
    wt = (torch.randn(N, K, device=gpu) * 0.1).bfloat16()
    y1 = torch.randn(M, N, device=gpu)
    flat = y1.view(-1)
    sp = torch.tensor([0.0, -0.0, 1e-7, -1e-7, 3e-6, -3.9e-6, 15.9, -15.9, 16.0, -16.0, 40.0, -1e30, float("inf"), float("-inf"), float("nan")], device=gpu)
    idx = torch.randperm(flat.numel(), device=gpu)[:sp.numel() * 3]
    flat[idx] = sp.repeat(3)
    y1 = y1.bfloat16()
    st = torch.cuda.current_stream(gpu).cuda_stream
    # the two calls
    dact = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
    _lib.check(L.slak_linear_nt(dz.data_ptr(), wt.data_ptr(), None, dact.data_ptr(), None, M, N, K, st), "linear_nt")
    ref = torch.empty_like(dact); dbr = torch.empty(N, device=gpu)
    ws, nb = block_ops._workspace(L.slak_gelu_bwd_workspace_bytes(M, N), gpu)
    _lib.check(L.slak_gelu_backward_bias(dact.data_ptr(), y1.data_ptr(), ref.data_ptr(), dbr.data_ptr(), M, N, ws.data_ptr(), nb, st), "gelu")
    # the one call
    dy1 = torch.full((M, N), float("nan"), device=gpu, dtype=torch.bfloat16); db = torch.full((N,), float("nan"), device=gpu)
    nb2 = int(L.slak_linear_nt_gelu_bwd_workspace_bytes(M, N, K))
    ws2 = torch.empty(nb2, dtype=torch.uint8, device=gpu)
    _lib.check(L.slak_linear_nt_gelu_bwd(dz.data_ptr(), wt.data_ptr(), y1.data_ptr(), dy1.data_ptr(), db.data_ptr(), M, N, K, ws2.data_ptr(), nb2, st), "fused")
    torch.cuda.synchronize()
    assert torch.equal(dy1.view(torch.int16), ref.view(torch.int16))
    colfin = torch.isfinite(dy1.double().sum(0))
    got, want = db.double()[colfin], dy1.double().sum(0)[colfin]
    assert (got - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item()) * max(1.0, M ** 0.5 / 30)
    assert (db.double()[colfin] - dbr.double()[colfin]).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()) * max(1.0, M ** 0.5 / 30)


@pytest.mark.parametrize("M", [32, 96, 6272, 40032, 401408])
def test_linear_nt_gelu_bwd_dt_adds_the_next_product(M, gpu):
    """slak_linear_nt_gelu_bwd_dt: dy1 and the bias gradient carry the bits of slak_linear_nt_gelu_bwd (special pre-activations included); dt = dy1 . W1
    agrees with the stand-alone slak_linear_nt on the stored dy1 up to the order of the fp32 additions (a bf16 rounding flips now and then) and with fp64."""
    from slak_amd import _lib
    L = _lib.lib()
    N, K = 384, 96
    assert L.slak_linear_nt_gelu_bwd_dt_supported(M, N, K) == 1
    torch.manual_seed(M + 1)
    dz = (torch.randn(M, K, device=gpu) * 0.5).bfloat16()
    wt = (torch.randn(N, K, device=gpu) * 0.1).bfloat16()
    w1t = (torch.randn(K, N, device=gpu) * 0.1).bfloat16()
    w1p = w1t.view(3, 32, 6, 4, 2, 8).permute(2, 3, 0, 4, 1, 5).contiguous()      # fragment-major: [pair][k-step][row tile][lane half][row][8 k]
    y1 = torch.randn(M, N, device=gpu)
    flat = y1.view(-1)
    sp = torch.tensor([0.0, -0.0, 1e-7, -1e-7, 3e-6, 15.9, -15.9, 16.0, -16.0, 40.0, -1e30], device=gpu)
    flat[torch.randperm(flat.numel(), device=gpu)[:sp.numel() * 3]] = sp.repeat(3)
    y1 = y1.bfloat16()
    st = torch.cuda.current_stream(gpu).cuda_stream
    nb = int(L.slak_linear_nt_gelu_bwd_workspace_bytes(M, N, K))
    ws = torch.empty(nb, dtype=torch.uint8, device=gpu)
    ref = torch.empty(M, N, device=gpu, dtype=torch.bfloat16); dbr = torch.empty(N, device=gpu)
    _lib.check(L.slak_linear_nt_gelu_bwd(dz.data_ptr(), wt.data_ptr(), y1.data_ptr(), ref.data_ptr(), dbr.data_ptr(), M, N, K, ws.data_ptr(), nb, st), "two")
    dtr = torch.empty(M, K, device=gpu, dtype=torch.bfloat16)
    _lib.check(L.slak_linear_nt(ref.data_ptr(), w1t.data_ptr(), None, dtr.data_ptr(), None, M, K, N, st), "linear_nt")
    for rep in range(2):                                       # twice: the second call runs over what the first left in LDS and the workspace
        dy1 = torch.full((M, N), float("nan"), device=gpu, dtype=torch.bfloat16); db = torch.full((N,), float("nan"), device=gpu)
        dt = torch.full((M, K), float("nan"), device=gpu, dtype=torch.bfloat16)
        _lib.check(L.slak_linear_nt_gelu_bwd_dt(dz.data_ptr(), wt.data_ptr(), y1.data_ptr(), w1p.data_ptr(), dy1.data_ptr(), dt.data_ptr(), db.data_ptr(),
                                                M, N, K, ws.data_ptr(), nb, st), "fused")
        torch.cuda.synchronize()
        assert torch.equal(dy1.view(torch.int16), ref.view(torch.int16)) and torch.equal(db, dbr)
        want = ref.double() @ w1t.double().t()
        scale = want.abs().max().item()
        assert (dt.double() - want).abs().max().item() <= 2.0 ** -8 * 1.05 * scale
        assert (dt.double() - dtr.double()).abs().max().item() <= 2.0 ** -7 * scale
        assert (dt.view(torch.int16) != dtr.view(torch.int16)).float().mean().item() < 0.02


def test_linear_nt_gelu_bwd_declines_what_it_does_not_cover(gpu):
    from slak_amd import _lib
    L = _lib.lib()
    for (M, N, K) in [(6272, 768, 192), (6272, 384, 192), (100, 384, 96), (6272, 512, 128), (0, 384, 96)]:
        assert L.slak_linear_nt_gelu_bwd_supported(M, N, K) == 0
        assert L.slak_linear_nt_gelu_bwd_workspace_bytes(M, N, K) == 0
    x = torch.zeros(100, 384, device=gpu, dtype=torch.bfloat16); db = torch.zeros(384, device=gpu)
    rc = L.slak_linear_nt_gelu_bwd(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), db.data_ptr(), 100, 384, 96, x.data_ptr(), 1 << 20, None)
    assert rc == _lib.ERR_UNSUPPORTED


@pytest.mark.parametrize("M", [32, 50, 6272, 40033, 401408])
def test_linear_mlp_fwd_is_pwconv1_gelu_pwconv2(M, gpu):
    """slak_linear_mlp_fwd (stage 1: C = 96): y1 and a bit for bit what slak_linear_nt(.., gelu_out) stores (row tails included); z against the
    fp64 product of the STORED a with w2 (+ b2) within half a bf16 ulp + fp32 accumulation noise, and equal to the two-launch z except where the
    fp32 sums differ in their last bits (a rare one-ulp flip of the bf16 rounding)."""
    from slak_amd import _lib
    L = _lib.lib()
    C, C4 = 96, 384
    assert L.slak_linear_mlp_fwd_supported(M, C, C4) == 1
    torch.manual_seed(M)
    x = torch.randn(M, C, device=gpu).bfloat16()
    w1 = (torch.randn(C4, C, device=gpu) * 0.1).bfloat16(); b1 = (torch.randn(C4, device=gpu) * 0.1).bfloat16()
    w2 = (torch.randn(C, C4, device=gpu) * 0.05).bfloat16(); b2 = (torch.randn(C, device=gpu) * 0.1).bfloat16()
    st = torch.cuda.current_stream(gpu).cuda_stream
    y1r = torch.empty(M, C4, device=gpu, dtype=torch.bfloat16); ar = torch.empty_like(y1r); zr = torch.empty(M, C, device=gpu, dtype=torch.bfloat16)
    _lib.check(L.slak_linear_nt(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), y1r.data_ptr(), ar.data_ptr(), M, C4, C, st), "nt1")
    _lib.check(L.slak_linear_nt(ar.data_ptr(), w2.data_ptr(), b2.data_ptr(), zr.data_ptr(), None, M, C, C4, st), "nt2")
    nan = float("nan")
    y1 = torch.full_like(y1r, nan); a = torch.full_like(ar, nan); z = torch.full_like(zr, nan)
    _lib.check(L.slak_linear_mlp_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y1.data_ptr(), a.data_ptr(), z.data_ptr(),
                                     M, C, C4, st), "mlp")
    torch.cuda.synchronize()
    assert torch.equal(y1, y1r) and torch.equal(a, ar)
    ref = a.double() @ w2.double().t() + b2.double()
    err = (z.double() - ref).abs()
    tol = ref.abs() * 2.0 ** -8 * 1.01 + 1e-4
    assert bool((err <= tol).all()), float((err / tol).max())
    differ = (z != zr)
    assert differ.float().mean().item() < 2e-3
    assert ((z.double() - zr.double()).abs() <= zr.double().abs() * 2.0 ** -7 + 1e-4).all()      # (one bf16 ulp; fp32 noise where z is ~0)


def test_linear_mlp_fwd_declines_what_it_does_not_cover(gpu):
    from slak_amd import _lib
    L = _lib.lib()
    for (M, C, C4) in [(6272, 192, 768), (6272, 128, 512), (6272, 96, 192), (0, 96, 384)]:
        assert L.slak_linear_mlp_fwd_supported(M, C, C4) == 0
    x = torch.zeros(64, 768, device=gpu, dtype=torch.bfloat16)
    assert L.slak_linear_mlp_fwd(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), None, x.data_ptr(), x.data_ptr(), x.data_ptr(), 64, 192, 768, None) == _lib.ERR_UNSUPPORTED
