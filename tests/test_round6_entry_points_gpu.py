"""Round-6 C-ABI additions (include/slak_hip.h, ABI 8), each against the entry point it extends (pytest -m gpu):
slak_bn3_backward_{local,apply}_to write the six BatchNorm parameter gradients through one pointer each -- the bits of the [3][C] forms;
slak_bn3_forward_sums_counted / slak_bn3_backward_sums_dup fold the fill / clone launches of the SyncBatchNorm exchange into the sums launch;
slak_pack_w1t_fragments is the layout contract of slak_linear_nt_gelu_bwd_dt's w1p operand (models/SLaK.py:38-47, 156-165 backwards)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _ptr3(ts):
    return (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])


@pytest.mark.parametrize("N,C,H,W", [(4, 96, 56, 56), (3, 384, 14, 14), (5, 768, 7, 7), (2, 36, 9, 10)])
def test_bn3_backward_with_one_destination_per_gradient(N, C, H, W, gpu):
    from slak_amd import _lib, block_ops
    L = _lib.lib()
    torch.manual_seed(C)
    P = H * W
    ys = [(torch.randn(N, C, H, W, device=gpu) * (1 + i) + 0.2 * i).bfloat16() for i in range(3)]
    dout = torch.randn(N, C, H, W, device=gpu).bfloat16()
    gs = [torch.rand(C, device=gpu) + 0.5 for _ in range(3)]
    stats = torch.stack([torch.stack([y.float().mean((0, 2, 3)), 1.0 / (y.float().var((0, 2, 3), unbiased=False) + 1e-5).sqrt()], 1) for y in ys], 1).reshape(C * 6).contiguous()
    nb = L.slak_bn3_workspace_bytes(N, C)
    ws = torch.empty(nb, dtype=torch.uint8, device=gpu)
    st = torch.cuda.current_stream().cuda_stream

    def outs():
        return torch.empty(C * 9, device=gpu), [torch.empty_like(y) for y in ys]
    # the [3][C] form
    bc0, d0 = outs()
    dg0 = torch.empty(3, C, device=gpu); db0 = torch.empty(3, C, device=gpu)
    _lib.check(L.slak_bn3_backward_local(dout.data_ptr(), *[y.data_ptr() for y in ys], stats.data_ptr(), _ptr3(gs), bc0.data_ptr(), dg0.data_ptr(),
                                         db0.data_ptr(), *[d.data_ptr() for d in d0], N, C, P, ws.data_ptr(), nb, st), "local")
    # one destination per gradient, scattered over a flat buffer in a bucket-like order (reverse, odd gaps that keep 16-byte alignment)
    flat = torch.full((6 * C + 64,), float("nan"), device=gpu)
    offs = [5 * C + 40, 4 * C + 32, 3 * C + 24, 2 * C + 16, C + 8, 0]
    dst = [flat[o:o + C] for o in offs]
    bc1, d1 = outs()
    _lib.check(L.slak_bn3_backward_local_to(dout.data_ptr(), *[y.data_ptr() for y in ys], stats.data_ptr(), _ptr3(gs), bc1.data_ptr(),
                                            _ptr3(dst[0::2]), _ptr3(dst[1::2]), *[d.data_ptr() for d in d1], N, C, P, ws.data_ptr(), nb, st), "local_to")
    for b in range(3):
        assert torch.equal(dst[2 * b], dg0[b]) and torch.equal(dst[2 * b + 1], db0[b])
        assert torch.equal(d1[b], d0[b])
    assert torch.equal(bc0, bc1)
    used = torch.zeros_like(flat, dtype=torch.bool)
    for o in offs:
        used[o:o + C] = True
    assert bool(torch.isnan(flat[~used]).all())                      # nothing written outside the six destinations
    # the exchange form: sums (+ duplicate) -> apply, [3][C] against per-gradient destinations
    ls0 = torch.empty(C * 4, device=gpu); ls1 = torch.empty(C * 4, device=gpu); cp = torch.empty(C * 4, device=gpu)
    _lib.check(L.slak_bn3_backward_sums(dout.data_ptr(), *[y.data_ptr() for y in ys], stats.data_ptr(), ls0.data_ptr(), N, C, P, ws.data_ptr(), nb, st), "sums")
    _lib.check(L.slak_bn3_backward_sums_dup(dout.data_ptr(), *[y.data_ptr() for y in ys], stats.data_ptr(), ls1.data_ptr(), cp.data_ptr(), N, C, P,
                                            ws.data_ptr(), nb, st), "sums_dup")
    assert torch.equal(ls0, ls1) and torch.equal(ls0, cp)
    gsum = ls0 * 2                                                    # "two ranks with the same batch"
    bc2, d2 = outs(); dg2 = torch.empty(3, C, device=gpu); db2 = torch.empty(3, C, device=gpu)
    _lib.check(L.slak_bn3_backward_apply(dout.data_ptr(), *[y.data_ptr() for y in ys], gsum.data_ptr(), ls0.data_ptr(), float(2 * N * P), None, stats.data_ptr(),
                                         _ptr3(gs), bc2.data_ptr(), dg2.data_ptr(), db2.data_ptr(), *[d.data_ptr() for d in d2], N, C, P, st), "apply")
    flat.fill_(float("nan"))
    bc3, d3 = outs()
    _lib.check(L.slak_bn3_backward_apply_to(dout.data_ptr(), *[y.data_ptr() for y in ys], gsum.data_ptr(), ls0.data_ptr(), float(2 * N * P), None, stats.data_ptr(),
                                            _ptr3(gs), bc3.data_ptr(), _ptr3(dst[0::2]), _ptr3(dst[1::2]), *[d.data_ptr() for d in d3], N, C, P, st), "apply_to")
    for b in range(3):
        assert torch.equal(dst[2 * b], dg2[b]) and torch.equal(dst[2 * b + 1], db2[b]) and torch.equal(d3[b], d2[b])
    assert L.slak_bn3_backward_local_to(dout.data_ptr(), *[y.data_ptr() for y in ys], stats.data_ptr(), _ptr3(gs), bc1.data_ptr(),
                                        None, _ptr3(dst[1::2]), *[d.data_ptr() for d in d1], N, C, P, ws.data_ptr(), nb, st) == 1    # SLAK_ERR_INVALID_ARG


@pytest.mark.parametrize("N,C,H,W", [(4, 96, 28, 28), (3, 384, 14, 14), (5, 768, 7, 7)])
def test_bn3_forward_sums_counted_writes_the_element_count(N, C, H, W, gpu):
    from slak_amd import _lib
    L = _lib.lib()
    torch.manual_seed(C + 1)
    ys = [(torch.randn(N, C, H, W, device=gpu) * (1 + i)).bfloat16() for i in range(3)]
    nb = L.slak_bn3_workspace_bytes(N, C)
    ws = torch.empty(nb, dtype=torch.uint8, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    a = torch.full((6 * C + 1,), -1.0, dtype=torch.float64, device=gpu); b = a.clone()
    _lib.check(L.slak_bn3_forward_sums(*[y.data_ptr() for y in ys], a.data_ptr(), N, C, H * W, ws.data_ptr(), nb, st, None, None, 0), "sums")
    _lib.check(L.slak_bn3_forward_sums_counted(*[y.data_ptr() for y in ys], b.data_ptr(), N, C, H * W, ws.data_ptr(), nb, st, None, None, 0), "counted")
    assert torch.equal(a[:6 * C], b[:6 * C])
    assert a[6 * C].item() == -1.0 and b[6 * C].item() == float(N * H * W)


def test_pack_w1t_fragments_is_the_documented_permutation(gpu):
    from slak_amd import _lib, block_ops
    L = _lib.lib()
    torch.manual_seed(3)
    w1t = torch.randn(96, 384, device=gpu).bfloat16()
    ref = w1t.view(3, 32, 6, 4, 2, 8).permute(2, 3, 0, 4, 1, 5).contiguous()     # [pair][k-step][row tile][lane half][row][8 k]
    out = torch.empty_like(w1t)
    _lib.check(L.slak_pack_w1t_fragments(w1t.data_ptr(), out.data_ptr(), 384, 96, torch.cuda.current_stream().cuda_stream), "pack")
    assert torch.equal(out.view(-1), ref.view(-1))
    assert torch.equal(block_ops.w1_fragments(w1t).view(-1), ref.view(-1))
    assert L.slak_pack_w1t_fragments(w1t.data_ptr(), out.data_ptr(), 256, 96, None) == 2                  # SLAK_ERR_UNSUPPORTED: only the stage-1 shape has the fused launch
