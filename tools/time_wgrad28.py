"""dev: the 49x5 weight gradient on the 28x28 stage (192 channels, N = 128) -- SLAK_MFMA_VWAVE=0 for the transposing kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops
dev = torch.device("cuda:0")
N, C, H, W = 128, 192, 28, 28
x = torch.randn(N, C, H, W, device=dev).bfloat16(); dy = torch.randn_like(x)
for kh, kw in ((49, 5), (5, 49)):
    w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
    for _ in range(5): ops.dwconv2d_backward_filter(dy, x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.dwconv2d_backward_filter(dy, x, w)
    e1.record(); torch.cuda.synchronize()
    print("vwave=%s wgs=%s  %dx%d wgrad 28x28: %.1f us" % (os.environ.get("SLAK_MFMA_VWAVE", "1"), os.environ.get("SLAK_VWAVE_WGS", "3"), kh, kw, e0.elapsed_time(e1) / 50 * 1e3))
