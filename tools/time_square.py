"""tools/time_square.py -- kernels without a 5-tap side (square kernels: the reference's test grid, --Decom False) in bf16: the chunked matrix-core
path (AUTO) against the exact VALU kernels (DIRECT), forward / data gradient / weight gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def ev(fn, reps=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (N, C, H, K) in ((128, 96, 56, 31), (128, 96, 56, 51), (128, 192, 28, 13), (64, 64, 32, 31), (64, 64, 64, 7)):
    x = torch.randn(N, C, H, H, device=dev).bfloat16(); dy = torch.randn_like(x); w = torch.randn(C, 1, K, K, device=dev) * 0.02
    row = []
    for algo, name in ((_lib.ALGO_AUTO, "auto"), (_lib.ALGO_DIRECT, "direct")):
        L.slak_set_conv_algo(algo)
        tf = ev(lambda: ops.dwconv2d_forward(x, w)); kf = L.slak_debug_last_kernel().decode()
        td = ev(lambda: ops.dwconv2d_backward_data(dy, w)); tw = ev(lambda: ops.dwconv2d_backward_filter(dy, x, w)); kw_ = L.slak_debug_last_kernel().decode()
        row.append("%s: fwd %7.1f dgrad %7.1f wgrad %8.1f us (%s, %s)" % (name, tf, td, tw, kf, kw_))
    L.slak_set_conv_algo(_lib.ALGO_AUTO)
    print("%2dx%-2d on %dx%dx%dx%d bf16 | %s | %s" % (K, K, N, C, H, H, row[0], row[1]), flush=True)
