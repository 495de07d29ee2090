"""Event-timed: slak_linear_nt(gelu) + slak_linear_nt vs slak_linear_mlp_fwd at stage 1 of SLaK-T (M = 128*56*56, C = 96)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
M, C, C4 = 128 * 56 * 56, 96, 384
x = torch.randn(M, C, device=dev).bfloat16(); w1 = (torch.randn(C4, C, device=dev) * 0.1).bfloat16(); b1 = torch.zeros(C4, device=dev).bfloat16()
w2 = (torch.randn(C, C4, device=dev) * 0.05).bfloat16(); b2 = torch.zeros(C, device=dev).bfloat16()
y1 = torch.empty(M, C4, device=dev, dtype=torch.bfloat16); a = torch.empty_like(y1); z = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
burn = torch.randn(4096, 4096, device=dev)
def two():
    L.slak_linear_nt(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), y1.data_ptr(), a.data_ptr(), M, C4, C, st)
    L.slak_linear_nt(a.data_ptr(), w2.data_ptr(), b2.data_ptr(), z.data_ptr(), None, M, C, C4, st)
def one():
    assert L.slak_linear_mlp_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y1.data_ptr(), a.data_ptr(), z.data_ptr(), M, C, C4, st) == 0
for name, fn in (("two calls", two), ("one call", one), ("two calls", two), ("one call", one)):
    for _ in range(3): fn()
    for _ in range(20): burn @ burn
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-10s %8.2f us" % (name, e0.elapsed_time(e1) * 1000 / 30), flush=True)
