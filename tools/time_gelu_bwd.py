"""Event-timed: slak_linear_nt + slak_gelu_backward_bias vs slak_linear_nt_gelu_bwd at stage 1 of SLaK-T (M = 128*56*56, K = 96, N = 384)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib, block_ops
dev = torch.device("cuda:0"); L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
M, N, K = 128 * 56 * 56, 384, 96
dz = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); wt = (torch.randn(N, K, device=dev) * 0.1).bfloat16()
y1 = torch.randn(M, N, device=dev).bfloat16(); dact = torch.empty_like(y1); dy1 = torch.empty_like(y1); db = torch.empty(N, device=dev)
ws, nb = block_ops._workspace(max(L.slak_gelu_bwd_workspace_bytes(M, N), L.slak_linear_nt_gelu_bwd_workspace_bytes(M, N, K)), dev)
burn = torch.randn(4096, 4096, device=dev)
def two():
    L.slak_linear_nt(dz.data_ptr(), wt.data_ptr(), None, dact.data_ptr(), None, M, N, K, st)
    L.slak_gelu_backward_bias(dact.data_ptr(), y1.data_ptr(), dy1.data_ptr(), db.data_ptr(), M, N, ws.data_ptr(), nb, st)
def one():
    assert L.slak_linear_nt_gelu_bwd(dz.data_ptr(), wt.data_ptr(), y1.data_ptr(), dy1.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), nb, st) == 0
for name, fn in (("two calls", two), ("one call", one), ("two calls", two), ("one call", one)):
    for _ in range(3): fn()
    for _ in range(20): burn @ burn
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-10s %8.2f us   (%.0f GB/s on %d MB)" % (name, e0.elapsed_time(e1) * 1000 / 30, (M * K * 2 + 2 * M * N * 2) / (e0.elapsed_time(e1) / 30) / 1e6, (M * K * 2 + 2 * M * N * 2) >> 20), flush=True)
