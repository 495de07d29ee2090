"""Dev: time every SLaK-T dw-conv kernel shape (bf16, N=128) through the C ABI."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops
dev = torch.device("cuda:0")
_R = int(os.environ.get("SLAK_TIME_ALL_REPS", "0"))
def ev(fn, reps=20, batches=5):
    if _R:
        reps, batches = _R, 1
    for _ in range(2 if _R else 20): fn()
    best = 1e30
    for _ in range(batches):                      # min over batches: robust against clock ramp / stray activity
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
tot = 0
for (C, H, K, blocks) in ((96, 56, 51, 3), (192, 28, 49, 3), (384, 14, 47, 9), (768, 7, 13, 3)):
    x = torch.randn(128, C, H, H, device=dev).bfloat16(); dy = torch.randn_like(x)
    for (kh, kw) in ((K, 5), (5, K), (5, 5)):
        w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
        tf = ev(lambda: ops.dwconv2d_forward(x, w)); td = ev(lambda: ops.dwconv2d_backward_data(dy, w)); tw = ev(lambda: ops.dwconv2d_backward_filter(dy, x, w))
        by = 2 * x.numel() * 2
        tot += (tf + td + tw) * blocks
        print("C%-3d %2dx%-2d k%2dx%-2d  fwd %6.1f us %5.0f GB/s | dgrad %6.1f us | wgrad %6.1f us %5.0f GB/s" % (C, H, H, kh, kw, tf, by / tf / 1e3, td, tw, by / tw / 1e3), flush=True)
print("dwconv per step: %.2f ms" % (tot / 1e3))
