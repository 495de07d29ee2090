"""Dev: one conv op at several batch sizes (for a rocprofv3 kernel trace: slope = per-plane cost, intercept = fixed cost)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops
dev = torch.device("cuda:0")
op, C, H, kh, kw = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
for N in (32, 64, 128, 256):
    x = torch.randn(N, C, H, H, device=dev).bfloat16(); dy = torch.randn_like(x); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
    for _ in range(8):
        if op == "wgrad": ops.dwconv2d_backward_filter(dy, x, w)
        elif op == "fwd": ops.dwconv2d_forward(x, w)
        else: ops.dwconv2d_backward_data(dy, w)
    torch.cuda.synchronize()
