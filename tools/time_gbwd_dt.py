"""Event-timed stage-1 backward data path: slak_linear_nt_gelu_bwd + slak_linear_nt (dt) against slak_linear_nt_gelu_bwd_dt (N = 128 x 56 x 56 rows)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib()
M, N, K = int(os.environ.get("BATCH", "128")) * 3136, 384, 96
dz = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); wt = (torch.randn(N, K, device=dev) * 0.1).bfloat16(); w1t = (torch.randn(K, N, device=dev) * 0.1).bfloat16()
w1p = w1t.view(3, 32, 6, 4, 2, 8).permute(2, 3, 0, 4, 1, 5).contiguous()
y1 = torch.randn(M, N, device=dev).bfloat16(); dy1 = torch.empty_like(y1); dt = torch.empty(M, K, device=dev, dtype=torch.bfloat16); db = torch.empty(N, device=dev)
nb = int(L.slak_linear_nt_gelu_bwd_workspace_bytes(M, N, K)); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


def two():
    L.slak_linear_nt_gelu_bwd(dz.data_ptr(), wt.data_ptr(), y1.data_ptr(), dy1.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), nb, st)
    L.slak_linear_nt(dy1.data_ptr(), w1t.data_ptr(), None, dt.data_ptr(), None, M, K, N, st)


def one():
    L.slak_linear_nt_gelu_bwd_dt(dz.data_ptr(), wt.data_ptr(), y1.data_ptr(), w1p.data_ptr(), dy1.data_ptr(), dt.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), nb, st)


print("two launches %.1f us   one launch %.1f us" % (timed(two), timed(one)))
print("gelu_bwd alone %.1f us" % timed(lambda: L.slak_linear_nt_gelu_bwd(dz.data_ptr(), wt.data_ptr(), y1.data_ptr(), dy1.data_ptr(), db.data_ptr(), M, N, K, ws.data_ptr(), nb, st)))
