"""dev: the LayerNorm kernels on the 14x14 stage (C = 384, N = 128), forward and forward+backward through autograd (host overhead included)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slak_amd import block_ops
dev = torch.device("cuda:0")
N, C, H = 128, 384, 14
x = torch.randn(N, C, H, H, device=dev).bfloat16().requires_grad_(True)
w = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
g = torch.randn(N, H, H, C, device=dev).bfloat16()
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
f = t(lambda: block_ops.ln_nchw_to_nhwc(x, w, b, 1e-6))
fb = t(lambda: block_ops.ln_nchw_to_nhwc(x, w, b, 1e-6).backward(g))
print("chan=%s ln fwd C384 14x14: %.1f us   fwd + bwd (autograd, reduce launch): %.1f us" % (os.environ.get("SLAK_RT_CHAN", "1"), f, fb))
