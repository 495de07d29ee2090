"""Dev: planes-per-group sweep of the DMA conv kernel (SLAK_DMA_G multiplies G)."""
import sys, os, subprocess
code = r'''
import sys, os
sys.path.insert(0, os.getcwd())
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
out = []
for (C, H, kh, kw) in ((96, 56, 5, 51), (96, 56, 51, 5), (96, 56, 5, 5), (192, 28, 5, 49), (192, 28, 49, 5)):
    x = torch.randn(128, C, H, H, device=dev).bfloat16(); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
    ref = None
    out.append("%dx%d k%dx%d %.1f" % (H, H, kh, kw, ev(lambda: ops.dwconv2d_forward(x, w))))
print(" ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("SLAK_")) + ": " + " | ".join(out))
'''
for g in sys.argv[2:]:
    env = dict(os.environ); env[sys.argv[1]] = g
    subprocess.run([sys.executable, "-c", code], env=env)
