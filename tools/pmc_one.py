"""Dev: run one conv shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops
dev = torch.device("cuda:0")
C, H, kh, kw = [int(a) for a in sys.argv[1:5]]
x = torch.randn(128, C, H, H, device=dev).bfloat16(); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
for _ in range(6): ops.dwconv2d_forward(x, w)
torch.cuda.synchronize()
