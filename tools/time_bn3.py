"""Event-timed bn3 backward (chansums + finalize + apply) at the four stages of SLaK-T; results vs SLAK_BN3_REF file if given (bit identity check across builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import block_ops
dev = torch.device("cuda:0")
burn = torch.randn(4096, 4096, device=dev)
torch.manual_seed(0)
outs = []
for (N, C, H) in [(128, 96, 56), (128, 192, 28), (128, 384, 14), (128, 768, 7)]:
    ys = [torch.randn(N, C, H, H, device=dev).bfloat16() for _ in range(3)]
    ds = torch.randn(N, C, H, H, device=dev).bfloat16()
    bns = [torch.nn.BatchNorm2d(C).to(dev) for _ in range(3)]
    gam = [bn.weight for bn in bns]; bet = [bn.bias for bn in bns]
    out, stats, count, count_dev = block_ops._bn3_forward_impl(ys[0], ys[1], ys[2], gam, bet, bns, None, None)
    fn = lambda: block_ops._bn3_backward_impl(ds, ys[0], ys[1], ys[2], gam, stats, None, count, count_dev)
    r = fn()
    outs.append([t.float().cpu() for t in r])
    for _ in range(3): fn()
    for _ in range(20): burn @ burn
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    print("%s bn3 backward %8.2f us (7 tensor passes + 4: %.0f GB/s)" % ((N, C, H), e0.elapsed_time(e1) * 1000 / 30, 11 * ds.numel() * 2 / (e0.elapsed_time(e1) / 30) / 1e6), flush=True)
ref = os.environ.get("SLAK_BN3_REF")
if ref and os.path.exists(ref):
    old = torch.load(ref)
    print("identical to", ref, all(torch.equal(a, b) for x, y in zip(old, outs) for a, b in zip(x, y)))
elif ref:
    torch.save(outs, ref)
