"""Dev: per-tap error map of the weight gradient (which taps are wrong)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from slak_amd import ops
dev = torch.device("cuda:0")
N, C, H, W, kh, kw = [int(a) for a in sys.argv[1:7]]
torch.manual_seed(0)
x = torch.randn(N, C, H, W, device=dev).bfloat16(); dy = torch.randn(N, C, H, W, device=dev).bfloat16()
w = torch.zeros(C, 1, kh, kw, device=dev)
dw = ops.dwconv2d_backward_filter(dy, x, w)
xr = x.float().requires_grad_(False); wr = w.clone().requires_grad_(True)
y = F.conv2d(xr, wr, None, 1, (kh // 2, kw // 2), 1, C)
y.backward(dy.float())
ref = wr.grad
err = (dw - ref).abs().amax(dim=(0, 1))          # [kh, kw]
scale = ref.abs().max().item()
print("max abs ref %.3f" % scale)
torch.set_printoptions(linewidth=200, precision=1, sci_mode=False)
print("rel err per tap (x1e3), rows = kh taps, cols = kw taps:")
print((err / scale * 1e3))
