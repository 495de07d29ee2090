import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from slak_amd import block_ops
dev = torch.device("cuda:0")
def ev(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (C, HW) in ((96, 56),):
    M = 128 * HW * HW
    t = torch.randn(M, C, device=dev).bfloat16(); w1 = torch.randn(4 * C, C, device=dev).bfloat16(); b1 = torch.randn(4 * C, device=dev).bfloat16()
    w2 = torch.randn(C, 4 * C, device=dev).bfloat16(); b2 = torch.randn(C, device=dev).bfloat16()
    a = torch.randn(M, 4 * C, device=dev).bfloat16(); dz = torch.randn(M, C, device=dev).bfloat16()
    w2t = w2.t().contiguous(); w1t = w1.t().contiguous()
    for name, fn, byt in (("fwd1+gelu skinny", lambda: block_ops.linear_nt(t, w1, b1, gelu=True), 2 * M * 9 * C),
                          ("fwd1+gelu library", lambda: F.gelu(F.linear(t, w1, b1)), 2 * M * 9 * C),
                          ("fwd2 skinny", lambda: block_ops.linear_nt(a, w2, b2), 2 * M * 5 * C),
                          ("fwd2 library", lambda: F.linear(a, w2, b2), 2 * M * 5 * C),
                          ("dact skinny", lambda: block_ops.linear_nt(dz, w2t), 2 * M * 5 * C),
                          ("dact library", lambda: torch.mm(dz, w2), 2 * M * 5 * C),
                          ("dt skinny", lambda: block_ops.linear_nt(a, w1t), 2 * M * 5 * C),
                          ("dt library", lambda: torch.mm(a, w1), 2 * M * 5 * C)):
        us = ev(fn)
        print("C=%d %-20s %7.1f us  %.2f TB/s" % (C, name, us, byt / us / 1e6))
