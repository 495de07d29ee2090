"""Dev check (GPU): MFMA depthwise-conv kernels vs the direct kernels and torch, plus timings.
    python tools/check_mfma.py [--time]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from slak_amd import ops, _lib

L = _lib.lib()
dev = torch.device("cuda:0")
SHAPES = [  # N, C, H, W, kh, kw
    (128, 96, 56, 56, 5, 51), (128, 96, 56, 56, 51, 5), (128, 96, 56, 56, 5, 5),
    (128, 192, 28, 28, 5, 49), (128, 192, 28, 28, 49, 5), (128, 192, 28, 28, 5, 5),
    (128, 384, 14, 14, 5, 47), (128, 384, 14, 14, 47, 5), (128, 384, 14, 14, 5, 5),
    (128, 768, 7, 7, 5, 13), (128, 768, 7, 7, 13, 5), (128, 768, 7, 7, 5, 5),
    (3, 5, 56, 56, 5, 51), (7, 3, 48, 40, 51, 5), (5, 4, 24, 28, 5, 31), (9, 3, 12, 16, 13, 5), (11, 2, 7, 8, 5, 9), (1, 1, 7, 7, 13, 5),
    (2, 3, 64, 64, 5, 51), (2, 3, 64, 64, 51, 5), (2, 3, 32, 32, 5, 31), (2, 3, 32, 32, 31, 5), (13, 3, 16, 16, 5, 7), (13, 3, 8, 8, 7, 5),
]
def ev(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
do_time = "--time" in sys.argv
bad = 0
for dt in (torch.bfloat16, torch.float16):
    for (N, C, H, W, kh, kw) in SHAPES:
        torch.manual_seed(N + H + kh)
        x = torch.randn(N, C, H, W, device=dev).to(dt)
        w = torch.randn(C, 1, kh, kw, device=dev) * 0.05
        wr = w.to(dt).float()
        ref = F.conv2d(x.float(), wr, None, 1, (kh // 2, kw // 2), 1, C)
        refd = F.conv_transpose2d(x.float(), wr, None, 1, (kh // 2, kw // 2), 0, C)
        L.slak_set_conv_algo(_lib.ALGO_MFMA)
        try:
            y = ops.dwconv2d_forward(x, w); dx = ops.dwconv2d_backward_data(x, w)
        except Exception as e:
            print("MFMA unsupported", (N, C, H, W, kh, kw), str(e)[:60]); L.slak_set_conv_algo(_lib.ALGO_AUTO); continue
        dy = torch.randn(N, C, H, W, device=dev).to(dt)
        dw = ops.dwconv2d_backward_filter(dy, x, w)
        dw2 = ops.dwconv2d_backward_filter(dy, x, w)
        L.slak_set_conv_algo(_lib.ALGO_DIRECT)
        dwref = ops.dwconv2d_backward_filter(dy, x, w)
        L.slak_set_conv_algo(_lib.ALGO_MFMA)
        e3 = (dw - dwref).abs().max().item() / max(1.0, dwref.abs().max().item())
        det = torch.equal(dw, dw2)
        e1 = (y.float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        e2 = (dx.float() - refd).abs().max().item() / max(1.0, refd.abs().max().item())
        msg = "%s %-28s fwd relerr %.2e dgrad relerr %.2e" % (str(dt)[6:], (N, C, H, W, kh, kw), e1, e2)
        msg += " wgrad relerr %.2e det %s" % (e3, det)
        if e1 > 6e-3 or e2 > 6e-3 or e3 > 2e-4 or not det: bad += 1; msg += "  <<<<<< BAD"
        if do_time and N == 128 and dt == torch.bfloat16:
            tm = ev(lambda: ops.dwconv2d_forward(x, w))
            L.slak_set_conv_algo(_lib.ALGO_DIRECT)
            td = ev(lambda: ops.dwconv2d_forward(x, w))
            by = 2 * x.numel() * 2
            tw_d = ev(lambda: ops.dwconv2d_backward_filter(dy, x, w))
            L.slak_set_conv_algo(_lib.ALGO_MFMA)
            tw_m = ev(lambda: ops.dwconv2d_backward_filter(dy, x, w))
            msg += "  fwd mfma %.1f us (%.0f GB/s) direct %.1f | wgrad mfma %.1f us (%.0f GB/s) direct %.1f" % (tm, by / tm / 1e3, td, tw_m, by / tw_m / 1e3, tw_d)
        L.slak_set_conv_algo(_lib.ALGO_AUTO)
        print(msg, flush=True)
print("BAD" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)
