"""dev: the forward residual kernel on the 14x14 stage (C = 384, N = 128) as a training step runs it (fp32 residual stream + bf16 copy)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slak_amd import block_ops
dev = torch.device("cuda:0")
N, C, H = 128, 384, 14
sc = torch.randn(N, C, H, H, device=dev); z = torch.randn(N, H, H, C, device=dev).bfloat16(); gamma = torch.randn(C, device=dev)
def run(): return block_ops._scale_residual_fwd(sc, z, gamma, None, True)
with torch.no_grad():
    r = run()
    ref = sc + (gamma * z.float()).permute(0, 3, 1, 2)
    print("max err vs torch:", (r[0] - ref).abs().max().item())
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print("chan=%s wide=%s sr fwd C384 14x14: %.1f us  (%.2f TB/s of 12 B/el, host overhead included)" % (os.environ.get("SLAK_RT_CHAN", "1"), os.environ.get("SLAK_RT_WIDE", "1"), us, N * C * H * H * 12 / us / 1e6))
