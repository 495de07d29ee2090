"""Time the block-tail kernels at the shapes of a bs-128 SLaK-T step (dev tool; run with SLAK_TAIL_REG=0/1 to A/B)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slak_amd import block_ops

def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

dev = torch.device("cuda:0")
print("SLAK_TAIL_REG =", os.environ.get("SLAK_TAIL_REG"))
for (N, C, H) in [(128, 96, 56), (128, 192, 28), (128, 384, 14), (128, 768, 7), (64, 128, 56), (64, 256, 28)]:
    x = torch.randn(N, C, H, H, device=dev).bfloat16().requires_grad_(True)
    w = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
    g = torch.randn(N, H, H, C, device=dev).bfloat16()
    el = N * C * H * H
    y = block_ops.ln_nchw_to_nhwc(x, w, b, 1e-6)
    f = t(lambda: block_ops.ln_nchw_to_nhwc(x, w, b, 1e-6))
    fb = t(lambda: block_ops.ln_nchw_to_nhwc(x, w, b, 1e-6).backward(g))
    sc = torch.randn(N, C, H, H, device=dev, requires_grad=True)
    z = torch.randn(N, H, H, C, device=dev).bfloat16().requires_grad_(True)
    gamma = torch.randn(C, device=dev, requires_grad=True)
    dout = torch.randn(N, C, H, H, device=dev)
    sf = t(lambda: block_ops.scale_residual(sc, z, gamma, None))
    sfb = t(lambda: block_ops.scale_residual(sc, z, gamma, None).backward(dout))
    d16 = dout.bfloat16()
    with torch.no_grad():                        # the variants a training step runs: bf16 copy of out; second gradient stream
        t(lambda: block_ops._scale_residual_fwd(sc, z, gamma, None, True), 10)
        t(lambda: block_ops._scale_residual_bwd(z, gamma, None, sc.dtype, dout, d16), 10)
    print("N%d C%d %dx%d: ln fwd %.1f us (%.2f TB/s of 4 B/el)  ln fwd+bwd %.1f us (autograd incl.)   sr fwd %.1f us (%.2f TB/s of 10 B/el)  sr fwd+bwd %.1f us"
          % (N, C, H, H, f, el * 4 / f / 1e6, fb, sf, el * 10 / sf / 1e6, sfb))
