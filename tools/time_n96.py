"""Event-timed slak_linear_nt with N = 96 (pwconv2 forward / dy1 . W1 at stage 1 of SLaK-T: M = 128*56*56, K = 384)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
burn = torch.randn(4096, 4096, device=dev)
for (M, K) in [(128 * 56 * 56, 384), (128 * 56 * 56, 96), (64 * 56 * 56, 192)]:
    x = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(96, K, device=dev) * 0.05).bfloat16(); b = torch.zeros(96, device=dev).bfloat16()
    y = torch.empty(M, 96, device=dev, dtype=torch.bfloat16)
    fn = lambda: L.slak_linear_nt(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None, M, 96, K, st)
    assert fn() == 0
    ref = (x[:4096].double() @ w.double().t()).float()
    err = (y[:4096].float() - ref).abs().max().item()
    for _ in range(3): fn()
    for _ in range(20): burn @ burn
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 30
    print("M=%d K=%d N=96: %8.2f us  %.0f GB/s  (max err vs fp64 on 4096 rows %.3g)" % (M, K, us, (M * K * 2 + M * 96 * 2) / us / 1e3, err), flush=True)
