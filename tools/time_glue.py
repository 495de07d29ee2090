"""Dev: time the block-tail and branch-BN kernels at the four SLaK-T stage shapes (N=128), report GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from slak_amd import block_ops
dev = torch.device("cuda:0")
def ev(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (C, H) in ((96, 56), (192, 28), (384, 14), (768, 7)):
    N = 128; S = N * C * H * H
    x = torch.randn(N, C, H, H, device=dev).bfloat16().requires_grad_(True)
    w = torch.ones(C, device=dev, requires_grad=True); b = torch.zeros(C, device=dev, requires_grad=True)
    y = block_ops.ln_nchw_to_nhwc(x, w, b, 1e-6); g = torch.randn_like(y)
    t_lnf = ev(lambda: block_ops.ln_nchw_to_nhwc(x, w, b, 1e-6))
    t_lnb = ev(lambda: torch.autograd.grad(y, (x, w, b), g, retain_graph=True))
    sc = torch.randn(N, C, H, H, device=dev).requires_grad_(True); z = torch.randn(N, H, H, C, device=dev).bfloat16().requires_grad_(True)
    gm = torch.ones(C, device=dev, requires_grad=True)
    o = block_ops.scale_residual(sc, z, gm, None); do = torch.randn_like(o)
    t_srf = ev(lambda: block_ops.scale_residual(sc, z, gm, None))
    t_srb = ev(lambda: torch.autograd.grad(o, (sc, z, gm), do, retain_graph=True))
    bns = [nn.BatchNorm2d(C).to(dev) for _ in range(3)]
    ys = [torch.randn(N, C, H, H, device=dev).bfloat16().requires_grad_(True) for _ in range(3)]
    ob = block_ops.branch_bn3(*ys, *bns); dob = torch.randn_like(ob)
    t_bnf = ev(lambda: block_ops.branch_bn3(*ys, *bns))
    t_bnb = ev(lambda: torch.autograd.grad(ob, ys + [bn.weight for bn in bns], dob, retain_graph=True))
    B = S * 2
    print("C%-3d %2dx%-2d  ln fwd %6.1f us (%4.0f GB/s)  ln bwd %6.1f us (%4.0f)  sres fwd %6.1f us (%4.0f)  sres bwd %6.1f us (%4.0f)  bn3 fwd %6.1f us (%4.0f)  bn3 bwd %6.1f us (%4.0f)" % (
        C, H, H, t_lnf, 2 * B / t_lnf / 1e3, t_lnb, 3 * B / t_lnb / 1e3, t_srf, 5 * B / t_srf / 1e3, t_srb, 4 * B / t_srb / 1e3,
        t_bnf, 7 * B / t_bnf / 1e3, t_bnb, 11 * B / t_bnb / 1e3), flush=True)
