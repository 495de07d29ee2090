"""dev: slak_gelu_backward_bias at the four stage shapes of a bs-128 SLaK-T step (event-timed; 6 bytes per element)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slak_amd import _lib, block_ops
dev = torch.device("cuda:0"); L = _lib.lib()
tot = 0.0
for (M, cols, blocks) in ((401408, 384, 3), (100352, 768, 3), (25088, 1536, 9), (6272, 3072, 3)):
    dact = torch.randn(M, cols, device=dev).bfloat16(); y1 = torch.randn(M, cols, device=dev).bfloat16(); dy1 = torch.empty_like(dact)
    db = torch.empty(cols, device=dev)
    ws, nb = block_ops._workspace(L.slak_gelu_bwd_workspace_bytes(M, cols), dev)
    st = torch.cuda.current_stream().cuda_stream
    def run(): _lib.check(L.slak_gelu_backward_bias(dact.data_ptr(), y1.data_ptr(), dy1.data_ptr(), db.data_ptr(), M, cols, ws.data_ptr(), nb, st), "gelu")
    for _ in range(5): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): run()
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    tot += us * blocks
    print("M %6d cols %4d: %6.1f us  %.2f TB/s" % (M, cols, us, 6.0 * M * cols / us / 1e6))
print("per step: %.3f ms" % (tot / 1e3))
