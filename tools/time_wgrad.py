"""Dev: time + check one weight-gradient shape.  python tools/time_wgrad.py N C H W kh kw [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from slak_amd import ops
dev = torch.device("cuda:0")
N, C, H, W, kh, kw = [int(a) for a in sys.argv[1:7]]
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 50
torch.manual_seed(0)
x = torch.randn(N, C, H, W, device=dev).bfloat16(); dy = torch.randn(N, C, H, W, device=dev).bfloat16()
w = torch.zeros(C, 1, kh, kw, device=dev)
dw = ops.dwconv2d_backward_filter(dy, x, w)
n0 = min(N, 8)
wr = w.clone().requires_grad_(True)
if N <= 16:
    F.conv2d(x.float(), wr, None, 1, (kh // 2, kw // 2), 1, C).backward(dy.float())
    print("max rel err %.2e" % ((dw - wr.grad).abs().max().item() / wr.grad.abs().max().item()))
for _ in range(5): ops.dwconv2d_backward_filter(dy, x, w)
torch.cuda.synchronize()
a, b = torch.cuda.Event(True), torch.cuda.Event(True)
a.record()
for _ in range(reps): ops.dwconv2d_backward_filter(dy, x, w)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / reps * 1e3
print("N%d C%d %dx%d k%dx%d: %.1f us  (%.0f GB/s algorithmic)  env NB=%s" % (N, C, H, W, kh, kw, us, 2 * x.numel() * 2 / us / 1e3, os.environ.get("SLAK_VROWS_NB")))
