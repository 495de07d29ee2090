"""dev: the backward residual kernel on the 14x14 stage (C = 384, N = 128) as a training step runs it (fp32 + bf16 gradient streams)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slak_amd import block_ops
dev = torch.device("cuda:0")
N, C, H = 128, 384, 14
z = torch.randn(N, H, H, C, device=dev).bfloat16(); gamma = torch.randn(C, device=dev)
dout = torch.randn(N, C, H, H, device=dev); d16 = dout.bfloat16()
def run(): return block_ops._scale_residual_bwd(z, gamma, None, torch.float32, dout, d16)
with torch.no_grad():
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("chan=%s sr bwd C384 14x14 (+ reduce launch, host overhead): %.1f us  (%.2f TB/s of 14 B/el)" % (os.environ.get("SLAK_RT_CHAN", "1"), us, N * C * H * H * 14 / us / 1e6))
