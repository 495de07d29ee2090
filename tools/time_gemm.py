"""dev: achieved HBM bandwidth of the library GEMMs of the block MLP (pwconv1/pwconv2 forward and the backward GEMMs) at the SLaK-T bs-128 shapes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def ev(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = 0
for (C, HW, blocks) in ((96, 56, 3), (192, 28, 3), (384, 14, 9), (768, 7, 3)):
    M = 128 * HW * HW
    t = torch.randn(M, C, device=dev).bfloat16(); w1 = torch.randn(4 * C, C, device=dev).bfloat16(); b1 = torch.randn(4 * C, device=dev).bfloat16()
    w2 = torch.randn(C, 4 * C, device=dev).bfloat16(); b2 = torch.randn(C, device=dev).bfloat16()
    y1 = F.linear(t, w1, b1); a = F.gelu(y1); dz = torch.randn(M, C, device=dev).bfloat16(); dy1 = torch.randn_like(y1)
    rows = []
    for name, fn, byt in (("fwd1 t@W1", lambda: F.linear(t, w1, b1), 2 * M * 5 * C),
                          ("gelu", lambda: F.gelu(y1), 2 * M * 8 * C),
                          ("fwd2 a@W2", lambda: F.linear(a, w2, b2), 2 * M * 5 * C),
                          ("dact dz@W2", lambda: torch.mm(dz, w2), 2 * M * 5 * C),
                          ("dt dy1@W1", lambda: torch.mm(dy1, w1), 2 * M * 5 * C)):
        us = ev(fn); tot += us * blocks
        rows.append("%s %.1f us %.2f TB/s" % (name, us, byt / us / 1e6))
    print("C=%d M=%d: " % (C, M) + " | ".join(rows))
print("sum over blocks: %.2f ms" % (tot / 1e3))
