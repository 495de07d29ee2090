"""Dev: ablation timings of the DMA conv kernel (SLAK_DMA_ABLATE=0..3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops
dev = torch.device("cuda:0")
def ev(fn, reps=20, batches=5):
    for _ in range(20): fn()
    best = 1e30
    for _ in range(batches):                      # min over batches: robust against clock ramp / stray activity
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): fn()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best
for (C, H, kh, kw) in ((96, 56, 5, 51), (96, 56, 51, 5), (192, 28, 5, 49), (192, 28, 49, 5)):
    x = torch.randn(128, C, H, H, device=dev).bfloat16(); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
    r = []
    for a in (0, 1, 2, 3):
        os.environ["SLAK_DMA_ABLATE"] = str(a)
        r.append(ev(lambda: ops.dwconv2d_forward(x, w)))
    os.environ["SLAK_DMA_ABLATE"] = "0"
    print("C%d %dx%d k%dx%d: full %.1f | no-mfma %.1f | no-dma %.1f | no-store %.1f us" % (C, H, H, kh, kw, *r))
