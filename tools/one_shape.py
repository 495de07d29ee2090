import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops, _lib
dev = torch.device("cuda:0")
for N in (6, 128):
    for (C, H, kh, kw) in ((96, 56, 5, 51), (96, 56, 51, 5), (192, 28, 5, 49), (192, 28, 49, 5)):
        x = torch.randn(N, C, H, H, device=dev).bfloat16(); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
        for _ in range(6):
            ops.dwconv2d_forward(x, w); ops.dwconv2d_backward_filter(x, x, w)
        torch.cuda.synchronize()
