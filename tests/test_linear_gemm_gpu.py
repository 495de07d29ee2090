"""slak_linear_gemm (csrc/linear_gemm.hip, round 5): the pointwise Linear layers of stages 2-4 (models/SLaK.py:156-165: pwconv1 -> nn.GELU() -> pwconv2 and
their data gradients) as ONE launch per GEMM with the elementwise neighbour in its epilogue, through the C ABI.

Checkers: the same product in fp64 on the bf16 operands (torch CPU), rounded once; nn.GELU() (exact erf) of the ROUNDED pre-activation as F.gelu of a bf16
tensor computes it; for the backward epilogue, operands whose products are EXACT in fp32 make the intermediate `dact` the same bits whatever the summation
order, so dy1 must then be BIT-IDENTICAL to slak_gelu_backward_bias applied to the library's dact (the two launches the kernel replaces)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

GELU, DGELU = 1, 2


def _call(a, b, bias, epi, y1=None):
    from slak_amd import _lib
    L = _lib.lib()
    M, K = a.shape
    N = b.shape[0]
    dev = a.device
    assert L.slak_linear_gemm_supported(M, N, K, epi) == 1, (M, N, K, epi)
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16)
    out2 = torch.full((M, N), float("nan"), device=dev, dtype=torch.bfloat16) if epi == GELU else None
    db = torch.full((N,), float("nan"), device=dev) if epi == DGELU else None
    nb = L.slak_linear_gemm_workspace_bytes(M, N, K, epi)
    ws = torch.empty(max(nb, 16), device=dev, dtype=torch.uint8)
    _lib.check(L.slak_linear_gemm(a.data_ptr(), b.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
                                  out2.data_ptr() if out2 is not None else None, y1.data_ptr() if y1 is not None else None,
                                  db.data_ptr() if db is not None else None, M, N, K, epi, ws.data_ptr() if nb else None, nb,
                                  torch.cuda.current_stream().cuda_stream), "slak_linear_gemm")
    torch.cuda.synchronize()
    return out, out2, db


SHAPES = [(25088, 1536, 384), (12544, 768, 192), (6272, 1536, 384), (129, 256, 192), (1, 256, 384), (127, 512, 192), (1000, 768, 384), (64 * 49, 1024, 384), (50176, 1024, 256), (333, 512, 256),
          (6272, 3072, 768), (200, 128, 768), (12544, 2048, 512), (129, 256, 512)]   # K = 512 / 768: two teams of four waves share K


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("with_bias", [True, False])
def test_gemm_with_gelu_epilogue_matches_fp64_product_and_torch_gelu(M, N, K, with_bias, gpu):
    torch.manual_seed(M + N + K)
    t = torch.randn(M, K, device=gpu).bfloat16()
    w = (torch.randn(N, K, device=gpu) * 0.07).bfloat16()
    bias = torch.randn(N, device=gpu).bfloat16() if with_bias else None
    y1, a, _ = _call(t, w, bias, GELU)
    assert not torch.isnan(y1.float()).any() and not torch.isnan(a.float()).any()          # every element was written (the outputs started as NaN)
    rows = torch.randperm(M, device=gpu)[:4096] if M > 4096 else torch.arange(M, device=gpu)
    ref = t[rows].double() @ w.double().t()
    if with_bias:
        ref = ref + bias.double()
    err = (y1[rows].double() - ref).abs()
    bound = 2.0 ** -8 * ref.abs() + 1e-5 * max(1.0, ref.abs().max().item())               # half a bf16 ulp + fp32 accumulation noise
    assert (err <= bound).all(), float((err - bound).max())
    want = torch.nn.functional.gelu(y1.float()).to(torch.bfloat16)                         # GELU of the ROUNDED pre-activation, rounded once
    d = (a.float() - want.float()).abs()
    assert (d <= 2.0 ** -7 * want.float().abs() + 1e-6).all(), d.max().item()             # at most one bf16 ulp (erf implementations differ in the last bit)
    assert (d == 0).float().mean().item() > 0.98
    y1b, ab, _ = _call(t, w, bias, GELU)                                                   # fixed summation order: run-to-run identical
    assert torch.equal(y1, y1b) and torch.equal(a, ab)


def _gelu_bwd_two_launches(dact, y1):
    from slak_amd import _lib
    L = _lib.lib()
    M, N = dact.shape
    nb = L.slak_gelu_bwd_workspace_bytes(M, N)
    ws = torch.empty(max(nb, 16), device=dact.device, dtype=torch.uint8)
    dy1 = torch.empty_like(dact)
    db = torch.empty(N, device=dact.device)
    _lib.check(L.slak_gelu_backward_bias(dact.data_ptr(), y1.data_ptr(), dy1.data_ptr(), db.data_ptr(), M, N, ws.data_ptr(), nb,
                                         torch.cuda.current_stream().cuda_stream), "slak_gelu_backward_bias")
    torch.cuda.synchronize()
    return dy1, db


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_with_gelu_backward_epilogue_is_the_two_launches_it_replaces(M, N, K, gpu):
    """Exact products: dz in {-2 .. 2}, W2 in {-1, 0, 1} / 8 -- every partial sum is a multiple of 1/8 below 2^11, exact in fp32 in any order, so the library's
    dact and the kernel's internal one are the same bits and dy1 must be IDENTICAL to at::mm followed by slak_gelu_backward_bias; y1 carries the special
    values (zeros, tiny, huge, infinities, NaN) on top of normal pre-activations."""
    torch.manual_seed(M * 3 + N + K)
    dz = torch.randint(-2, 3, (M, K), device=gpu).float().bfloat16()
    w2 = (torch.randint(-1, 2, (K, N), device=gpu).float() / 8).bfloat16()                # nn.Linear weight of pwconv2: [C][4C] = [K][N]
    y1 = (torch.randn(M, N, device=gpu) * 1.5).bfloat16()
    flat = y1.view(-1)
    special = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 2.0 ** -18, -2.0 ** -18, 15.9, -15.9, 16.0, -16.0, 300.0, -300.0, float("inf"), float("-inf"), float("nan")],
                           device=gpu).bfloat16()
    pos = torch.randperm(flat.numel(), device=gpu)[:special.numel() * 3]
    flat[pos] = special.repeat(3)
    dy1, _, db1 = _call(dz, w2.t().contiguous(), None, DGELU, y1=y1)
    dact = torch.mm(dz, w2)                                                                # the library's GEMM: exact sums, rounded once
    want, want_db = _gelu_bwd_two_launches(dact, y1)
    same = (dy1.view(torch.int16) == want.view(torch.int16)) | (torch.isnan(dy1.float()) & torch.isnan(want.float()))
    assert same.all(), int((~same).sum())
    # the bias gradient: column sums of the ROUNDED dy1 (fp32, another fixed order than the stand-alone kernel's)
    col = torch.nan_to_num(dy1.double(), nan=0.0, posinf=0.0, neginf=0.0).sum(0)
    fin = torch.isfinite(db1)
    scale = torch.nan_to_num(dy1.double(), nan=0.0, posinf=0.0, neginf=0.0).abs().sum(0)
    assert ((db1.double() - col).abs()[fin] <= 1e-5 * scale[fin] + 1e-6).all()
    bad_cols = ~torch.isfinite(dy1.float()).all(0)                                         # a column that holds inf / NaN sums to a non-finite value, as dy1.sum(0) does
    assert (~fin == bad_cols).all()
    assert torch.allclose(db1[fin], want_db[fin], rtol=1e-4, atol=1e-4 * float(scale.max()) / max(1, M) ** 0.5 + 1e-6)
    dy1b, _, db1b = _call(dz, w2.t().contiguous(), None, DGELU, y1=y1)                     # bitwise reproducible
    assert torch.equal(dy1.view(torch.int16), dy1b.view(torch.int16)) and torch.equal(db1.view(torch.int32), db1b.view(torch.int32))


@pytest.mark.parametrize("M,N,K", [(25088, 1536, 384), (777, 768, 192)])
def test_gemm_with_gelu_backward_epilogue_on_random_operands_vs_fp64(M, N, K, gpu):
    torch.manual_seed(5)
    dz = torch.randn(M, K, device=gpu).bfloat16()
    w2t = (torch.randn(N, K, device=gpu) * 0.05).bfloat16()
    y1 = torch.randn(M, N, device=gpu).bfloat16()
    dy1, _, db1 = _call(dz, w2t, None, DGELU, y1=y1)
    rows = torch.randperm(M, device=gpu)[:2048] if M > 2048 else torch.arange(M, device=gpu)
    dact = dz[rows].double() @ w2t.double().t()
    yy = y1[rows].double()
    gp = 0.5 * (1 + torch.erf(yy / math.sqrt(2))) + yy * torch.exp(-0.5 * yy * yy) / math.sqrt(2 * math.pi)
    want = dact * gp
    err = (dy1[rows].double() - want).abs()
    bound = 2.0 ** -7 * want.abs() + 2.0 ** -8 * dact.abs() + 1e-6                         # two roundings (dact, then the product)
    assert (err <= bound).all(), float((err - bound).max())
    col = dy1.double().sum(0)
    assert ((db1.double() - col).abs() <= 1e-5 * dy1.double().abs().sum(0) + 1e-6).all()


def test_what_the_kernel_does_not_cover_is_declined(gpu):
    from slak_amd import _lib
    L = _lib.lib()
    assert L.slak_linear_gemm_supported(3136, 4096, 1024, GELU) == 0                       # K = 1024 (SLaK-B stage 4: a K-half's B fragments would need 128 registers): library
    assert L.slak_linear_gemm_supported(25088, 384, 1536, 0) == 0                          # plain GEMMs with a long K stay with the library
    assert L.slak_linear_gemm_supported(1000, 384, 192, GELU) == 0                         # N not a multiple of 256
    assert L.slak_linear_gemm_supported(401408, 384, 96, GELU) == 0                        # stage 1 has its own streaming kernels (linear_skinny.hip)
    t = torch.zeros(16, 1024, device=gpu, dtype=torch.bfloat16)
    w = torch.zeros(4096, 1024, device=gpu, dtype=torch.bfloat16)
    o = torch.zeros(16, 4096, device=gpu, dtype=torch.bfloat16)
    rc = L.slak_linear_gemm(t.data_ptr(), w.data_ptr(), None, o.data_ptr(), o.data_ptr(), None, None, 16, 4096, 1024, GELU, None, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 2                                                                          # SLAK_ERR_UNSUPPORTED


def test_mlp_through_the_fused_gemms_matches_the_library_path(gpu):
    """The block MLP's autograd node on a stage-3-sized activation with the fused GEMMs and with SLAK_LINEAR_GEMM off (library GEMM + elementwise pass): forward and
    every gradient within bf16 rounding of each other; pwconv1's bias gradient to fp32 accuracy of a column sum."""
    from slak_amd import block_ops
    torch.manual_seed(3)
    C, M = 384, 8 * 14 * 14
    t = torch.randn(M, C, device=gpu).bfloat16()
    w1 = (torch.randn(4 * C, C, device=gpu) * 0.05).requires_grad_(True); b1 = torch.randn(4 * C, device=gpu).requires_grad_(True)
    w2 = (torch.randn(C, 4 * C, device=gpu) * 0.05).requires_grad_(True); b2 = torch.randn(C, device=gpu).requires_grad_(True)
    dz = torch.randn(M, C, device=gpu).bfloat16()
    res = {}
    saved = block_ops.use_linear_gemm
    try:
        for mode in (True, False):
            block_ops.use_linear_gemm = mode
            ti = t.clone().requires_grad_(True)
            for p in (w1, b1, w2, b2):
                p.grad = None
            z = block_ops.mlp_splitk(ti, w1, b1, w2, b2)
            z.backward(dz)
            res[mode] = [z.detach().float(), ti.grad.float()] + [p.grad.float().clone() for p in (w1, b1, w2, b2)]
    finally:
        block_ops.use_linear_gemm = saved
    for got, want, name in zip(res[True], res[False], ("z", "dt", "dw1", "db1", "dw2", "db2")):
        tol = 2e-2 * want.abs().max().item()
        assert (got - want).abs().max().item() <= tol, (name, (got - want).abs().max().item(), tol)


@pytest.mark.parametrize("shapes", [[(1536, 384)], [(384, 1536), (768, 192), (8, 8), (72, 200), (3072, 768), (96, 384)]])
def test_batched_transpose_is_t_contiguous(shapes, gpu):
    import ctypes
    from slak_amd import _lib
    torch.manual_seed(1)
    srcs = [torch.randn(r, c, device=gpu).bfloat16() for r, c in shapes]
    dsts = [torch.full((c, r), float("nan"), device=gpu, dtype=torch.bfloat16) for r, c in shapes]
    n = len(shapes)
    _lib.check(_lib.lib().slak_transpose_bf16_batch((ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs]), (ctypes.c_void_p * n)(*[t.data_ptr() for t in dsts]),
                                                    (ctypes.c_int * n)(*[r for r, _ in shapes]), (ctypes.c_int * n)(*[c for _, c in shapes]), n,
                                                    torch.cuda.current_stream().cuda_stream), "slak_transpose_bf16_batch")
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(d, s.t().contiguous())


def test_cached_transposed_weights_follow_the_parameter(gpu):
    """block_ops.lowp_param_t: the transposed bf16 copy the data-gradient GEMMs read is made once per parameter VERSION (one launch for all stale ones) and
    follows in-place updates of the master weight."""
    from slak_amd import block_ops
    saved = block_ops.cache_lowp_weights
    block_ops.cache_lowp_weights = True
    try:
        ws = [torch.nn.Parameter(torch.randn(r, c, device=gpu)) for r, c in ((1536, 384), (384, 1536), (768, 192))]
        ts = [block_ops.lowp_param_t(w) for w in ws]
        for w, t in zip(ws, ts):
            assert torch.equal(t, w.detach().bfloat16().t().contiguous())
        assert block_ops.lowp_param_t(ws[0]) is ts[0]                                      # same version: the cached tensor itself
        with torch.no_grad():
            for w in ws:
                w.mul_(1.5)                                                                # (bumps the version counter, as optimizer.step() does)
        t2 = [block_ops.lowp_param_t(w) for w in ws]
        for w, t, told in zip(ws, t2, ts):
            assert t is told and torch.equal(t, w.detach().bfloat16().t().contiguous())    # refreshed in place
    finally:
        block_ops.cache_lowp_weights = saved


@pytest.mark.parametrize("M,N,K", [(25088, 1536, 384), (6272, 3072, 768), (12544, 768, 192)])
def test_counted_waits_hold_while_another_stream_hammers_hbm(M, N, K, gpu):
    """The kernel waits for its LDS-DMA chunks with COUNTED vmcnt across tiles (operations retire in issue order) and reuses ring stages and staging tiles on that
    basis.  40 launches per epilogue of the same inputs while a second stream copies buffers of changing size (memory latencies move around): every output bit must equal
    the first launch's (tools/stress_linear_gemm.py is the long version: 1800 launches)."""
    from slak_amd import _lib
    L = _lib.lib()
    torch.manual_seed(K)
    t = torch.randn(M, K, device=gpu).bfloat16()
    w = (torch.randn(N, K, device=gpu) * 0.05).bfloat16()
    b = torch.randn(N, device=gpu).bfloat16()
    y1 = torch.randn(M, N, device=gpu).bfloat16()
    side = torch.cuda.Stream()
    junk = [torch.empty(n, device=gpu, dtype=torch.uint8) for n in (1 << 20, 29 << 20, 211 << 20)]
    junk2 = [torch.empty_like(j) for j in junk]
    for epi in (GELU, DGELU):
        first = _call(t, w, b if epi == GELU else None, epi, y1=y1 if epi == DGELU else None)
        for it in range(40):
            with torch.cuda.stream(side):
                junk2[it % 3].copy_(junk[it % 3])
            got = _call(t, w, b if epi == GELU else None, epi, y1=y1 if epi == DGELU else None)
            for a_, b_ in zip(first, got):
                if a_ is not None:
                    assert torch.equal(a_.view(torch.int16 if a_.dtype == torch.bfloat16 else torch.int32), b_.view(torch.int16 if b_.dtype == torch.bfloat16 else torch.int32)), (epi, it)
    torch.cuda.synchronize()
