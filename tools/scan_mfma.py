"""Dev: kernel time vs planes per workgroup (separates the per-workgroup prologue from the per-plane cost)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def ev(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
L.slak_set_conv_algo(_lib.ALGO_MFMA)
for (C, H, kh, kw) in ((96, 56, 5, 51), (96, 56, 51, 5), (192, 28, 5, 49), (192, 28, 49, 5)):
    for N in (6, 12, 24, 48, 96, 128, 192):
        x = torch.randn(N, C, H, H, device=dev).bfloat16(); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
        t = ev(lambda: ops.dwconv2d_forward(x, w))
        tw = ev(lambda: ops.dwconv2d_backward_filter(x, x, w))
        print("C%d %dx%d k%dx%d N=%3d  fwd %.1f us   wgrad %.1f us" % (C, H, H, kh, kw, N, t, tw), flush=True)
