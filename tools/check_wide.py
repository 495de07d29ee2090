"""dev: parity + timing of the wide-map (Wt > 64) MFMA kernels against the CPU oracle.  python tools/check_wide.py [time]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from slak_amd import ops, _lib

dev = torch.device("cuda:0")
L = _lib.lib()
L.slak_set_conv_algo(_lib.ALGO_MFMA)
SHAPES = [(3, 2, 96, 96, 5, 61), (3, 2, 96, 96, 61, 5), (2, 3, 96, 96, 5, 5), (2, 2, 128, 128, 5, 61), (2, 2, 128, 128, 61, 5),
          (5, 3, 80, 96, 5, 51), (5, 3, 96, 80, 51, 5), (2, 2, 112, 72, 57, 5), (2, 2, 72, 112, 5, 57), (1, 1, 96, 96, 61, 5), (4, 2, 128, 96, 5, 31)]
bad = 0
for (N, C, H, W, kh, kw) in SHAPES:
    for dtype in (torch.bfloat16, torch.float16):
        torch.manual_seed(N * 1000 + H * 10 + kh)
        x = torch.randn(N, C, H, W, device=dev).to(dtype); dy = torch.randn(N, C, H, W, device=dev).to(dtype)
        w = torch.randn(C, 1, kh, kw, device=dev) * 0.05
        xr, dyr, wr = (t.to(dtype).float().cpu().numpy() for t in (x, dy, w))
        for what, fn, ref in (("fwd", lambda: ops.dwconv2d_forward(x, w), lambda: oracle.dwconv2d_fwd(xr, wr)),
                              ("dgrad", lambda: ops.dwconv2d_backward_data(dy, w), lambda: oracle.dwconv2d_bwd_data(dyr, wr)),
                              ("wgrad", lambda: ops.dwconv2d_backward_filter(dy, x, w), lambda: oracle.dwconv2d_bwd_filter(dyr, xr, kh, kw))):
            try:
                got = fn().double().cpu().numpy(); r = ref()
                err = np.abs(got - r); scale = max(1.0, np.abs(r).max())
                tol = 1e-2 if what != "wgrad" else 1e-5 * max(1.0, (N * H * W) ** 0.5 / 30)
                ok = err.max() <= tol * scale
                if not ok:
                    bad += 1
                    idx = np.unravel_index(err.argmax(), err.shape)
                    nbad = int((err > tol * scale).sum())
                    print("FAIL", (N, C, H, W, kh, kw), dtype, what, "err %.3e scale %.3e at %s, %d bad of %d" % (err.max(), scale, idx, nbad, err.size))
                    if what != "wgrad":
                        e2 = (err > tol * scale)
                        print("   bad rows:", sorted(set(np.nonzero(e2)[2].tolist()))[:40], "cols:", sorted(set(np.nonzero(e2)[3].tolist()))[:40])
                else:
                    print("ok  ", (N, C, H, W, kh, kw), str(dtype)[6:], what, "%.2e" % err.max())
            except Exception as e:
                bad += 1
                print("EXC ", (N, C, H, W, kh, kw), dtype, what, repr(e)[:200])
print("failures:", bad)
if len(sys.argv) > 1:
    for (N, C, H, W, kh, kw) in [(64, 96, 96, 96, 61, 5), (64, 96, 96, 96, 5, 61), (64, 96, 96, 96, 5, 5), (16, 96, 128, 128, 61, 5), (16, 96, 128, 128, 5, 61),
                                 (64, 192, 48, 48, 59, 5), (64, 192, 48, 48, 5, 59), (64, 384, 24, 24, 57, 5), (64, 384, 24, 24, 5, 57), (64, 768, 12, 12, 13, 5), (64, 768, 12, 12, 5, 13)]:
        x = torch.randn(N, C, H, W, device=dev).bfloat16(); dy = torch.randn(N, C, H, W, device=dev).bfloat16()
        w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
        for what, fn in (("fwd", lambda: ops.dwconv2d_forward(x, w)), ("dgrad", lambda: ops.dwconv2d_backward_data(dy, w)), ("wgrad", lambda: ops.dwconv2d_backward_filter(dy, x, w))):
            try:
                for _ in range(3): fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): fn()
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / 20 * 1e3
                byt = 2 * N * C * H * W * 2
                print("time", (N, C, H, W, kh, kw), what, "%.1f us  %.2f TB/s  frac %.3f" % (us, byt / us / 1e6, byt / us / 1e6 / 8.0))
            except Exception as e:
                print("time EXC", (N, C, H, W, kh, kw), what, repr(e)[:200])
