"""tools/time_fp32.py -- fp32 dw-conv ops at the SLaK-T shapes (N = 128): exact VALU kernels vs the opt-in two-term split on the matrix cores."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops, _lib
dev = torch.device("cuda:0")
def ev(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
tot = {False: 0.0, True: 0.0}
for (C, H, K, blocks) in ((96, 56, 51, 3), (192, 28, 49, 3), (384, 14, 47, 9), (768, 7, 13, 3)):
    x = torch.randn(128, C, H, H, device=dev); dy = torch.randn_like(x)
    for (kh, kw) in ((K, 5), (5, K), (5, 5)):
        w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
        row = []
        for allow in (False, True):
            ops.allow_fp32_matrix_cores(allow)
            tf = ev(lambda: ops.dwconv2d_forward(x, w)); kf = _lib.lib().slak_debug_last_kernel().decode()
            td = ev(lambda: ops.dwconv2d_backward_data(dy, w)); tw = ev(lambda: ops.dwconv2d_backward_filter(dy, x, w)); kw_ = _lib.lib().slak_debug_last_kernel().decode()
            tot[allow] += (tf + td + tw) * blocks
            row.append("%s fwd %7.1f dgrad %7.1f wgrad %7.1f us (%s, %s)" % ("split" if allow else "exact", tf, td, tw, kf, kw_))
        print("C%-3d %2dx%-2d k%2dx%-2d | %s | %s" % (C, H, H, kh, kw, row[0], row[1]), flush=True)
ops.allow_fp32_matrix_cores(False)
print("fp32 dw-conv ops per SLaK-T step (per-branch launches): exact %.2f ms, split %.2f ms" % (tot[False] / 1e3, tot[True] / 1e3))
