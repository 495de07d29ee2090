"""dev: parity + timing of the one-launch three-branch kernels (56x56 / 28x28 class) against the per-branch kernels and the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from slak_amd import ops, _lib, block_ops
dev = torch.device("cuda:0")
L = _lib.lib()
bad = 0
for (N, C, H, W, K) in [(6, 5, 7, 7, 13), (17, 6, 7, 7, 13), (1, 1, 7, 7, 13), (4, 3, 6, 6, 9), (3, 2, 7, 5, 7), (5, 2, 5, 7, 9), (2, 130, 7, 7, 13), (5, 7, 14, 14, 47), (3, 4, 4, 4, 7), (2, 3, 7, 6, 9), (3, 2, 5, 5, 7), (40, 5, 7, 7, 13), (65, 3, 8, 8, 9), (33, 2, 3, 4, 7), (100, 9, 7, 7, 13),  (9, 2, 28, 28, 49), (2, 2, 48, 40, 31), (2, 3, 64, 64, 61), (3, 2, 32, 32, 31), (6, 2, 24, 24, 13), (7, 2, 28, 20, 13), (1, 1, 56, 56, 51), (11, 1, 20, 28, 49), (17, 2, 56, 56, 51), (33, 1, 28, 28, 49)]:
    for dtype in (torch.bfloat16, torch.float16):
        kind = L.slak_dwconv2d_tri_supported(_lib.SLAK_BF16 if dtype == torch.bfloat16 else _lib.SLAK_F16, N, C, H, W, K)
        torch.manual_seed(N + K)
        x = torch.randn(N, C, H, W, device=dev).to(dtype).requires_grad_(True)
        ws = [(torch.randn(C, 1, kh, kw, device=dev) * 0.05).requires_grad_(True) for kh, kw in ((K, 5), (5, K), (5, 5))]
        dys = [torch.randn(N, C, H, W, device=dev).to(dtype) for _ in range(3)]
        try:
            ys = block_ops.tri_dwconv(x, *ws)
            torch.autograd.backward(ys, dys)
            torch.cuda.synchronize()
        except Exception as e:
            print("EXC", (N, C, H, W, K), dtype, kind, repr(e)[:200]); bad += 1; continue
        msgs = []
        for bi, (y, w) in enumerate(zip(ys, ws)):
            yr = ops.dwconv2d_forward(x.detach(), w.detach())
            if not torch.equal(y, yr):
                d = (y.float() - yr.float()).abs()
                nb = int((d > 0).sum()); idx = np.unravel_index(int(d.argmax()), d.shape)
                msgs.append("fwd[%d] differs: max %.3e, %d elems, at %s" % (bi, d.max().item(), nb, idx))
        parts = [oracle.dwconv2d_bwd_data(dy.float().cpu().numpy(), w.detach().to(dtype).float().cpu().numpy()) for dy, w in zip(dys, ws)]
        ref = sum(parts); got = x.grad.double().cpu().numpy()
        ulp = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
        bound = ulp * (sum(np.abs(q) for q in parts) + np.abs(ref)) + 1e-5 * max(1.0, np.abs(ref).max())
        err = np.abs(got - ref)
        if not (err <= bound).all():
            e2 = err > bound
            msgs.append("dgrad: %d bad, max err %.3e; rows %s cols %s planes %s" % (int(e2.sum()), err.max(), sorted(set(np.nonzero(e2)[2].tolist()))[:20], sorted(set(np.nonzero(e2)[3].tolist()))[:20], sorted(set(zip(np.nonzero(e2)[0].tolist(), np.nonzero(e2)[1].tolist())))[:8]))
        if msgs:
            bad += 1; print("FAIL", (N, C, H, W, K), str(dtype)[6:], "kind", kind, "; ".join(msgs))
        else:
            print("ok  ", (N, C, H, W, K), str(dtype)[6:], "kind", kind)
print("failures:", bad)
if len(sys.argv) > 1:
    st = torch.cuda.current_stream(dev).cuda_stream
    for (N, C, H, W, K) in [(128, 768, 7, 7, 13), (64, 768, 12, 12, 13), (128, 96, 56, 56, 51), (128, 192, 28, 28, 49), (128, 384, 14, 14, 47), (64, 192, 48, 48, 59), (64, 384, 24, 24, 57)]:
        x = torch.randn(N, C, H, W, device=dev).bfloat16(); dys = [torch.randn_like(x) for _ in range(3)]
        ws = [torch.randn(C, 1, kh, kw, device=dev) * 0.02 for kh, kw in ((K, 5), (5, K), (5, 5))]
        ys = [torch.empty_like(x) for _ in range(3)]; dx = torch.empty_like(x)
        def tf(): _lib.check(L.slak_dwconv2d_tri_forward(x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, st))
        def td(): _lib.check(L.slak_dwconv2d_tri_backward_data(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), dx.data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, st))
        def sf():
            for w in ws: ops.dwconv2d_forward(x, w)
        def sd():
            for d, w in zip(dys, ws): ops.dwconv2d_backward_data(d, w)
        for what, fn in (("tri fwd", tf), ("tri dgrad", td), ("3x fwd", sf), ("3x dgrad", sd)):
            for _ in range(3): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            byt = 3 * 2 * N * C * H * W * 2
            print("time", (N, C, H, W, K), what, "%.1f us; per-op alg bytes %.0f MB -> %.2f TB/s (frac %.3f)" % (us, byt / 1e6, byt / us / 1e6, byt / us / 1e6 / 8))
