"""Dev: 7x7 kernels through the C ABI under rocprof (true durations): python tools/time_small.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(128, 768, 7, 7, device=dev).bfloat16(); dy = torch.randn_like(x)
for (kh, kw) in ((13, 5), (5, 13), (5, 5)):
    w = torch.randn(768, 1, kh, kw, device=dev) * 0.02
    for _ in range(20):
        ops.dwconv2d_forward(x, w); ops.dwconv2d_backward_data(dy, w); ops.dwconv2d_backward_filter(dy, x, w)
torch.cuda.synchronize()
