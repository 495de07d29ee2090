"""Dev: per-workgroup real-time stamps (100 MHz) of the weight-gradient DMA kernel."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
L.slak_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
names = ["zero+plan", "loop start", "loop end", "diag sums", "partial", "finish"]
order = [3, 1, 2, 4, 5, 6]
for (C, H, kh, kw) in ((96, 56, 5, 51), (96, 56, 51, 5), (192, 28, 5, 49)):
    x = torch.randn(128, C, H, H, device=dev).bfloat16(); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
    for _ in range(3): ops.dwconv2d_backward_filter(x, x, w)
    dbg = torch.zeros(64 + 8 * 4096, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    L.slak_debug_set_phase_buffer(dbg.data_ptr())
    ops.dwconv2d_backward_filter(x, x, w); torch.cuda.synchronize()
    L.slak_debug_set_phase_buffer(None)
    d = dbg[64:].cpu().view(-1, 8); d = d[d[:, 0] > 0]
    t0 = d[:, 0].min().item()
    print("C%d %dx%d k%dx%d: %d WGs; stamps in us since the first entry: median (min..max)" % (C, H, H, kh, kw, d.shape[0]))
    print("   entry %.1f (..%.1f)" % (((d[:, 0] - t0).float() / 100).median(), ((d[:, 0] - t0).float() / 100).max()))
    for n, k in zip(names, order):
        v = (d[:, k] - t0).float() / 100
        print("   %-10s %.1f (%.1f..%.1f)" % (n, v.median(), v.min(), v.max()))
