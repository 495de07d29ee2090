"""Dev: event-timed K x 5 + 5 x 5 pair weight gradient (and the 5 x K one) at a bench shape; env switches are read by the library at first use,
so A/B variants run as separate processes (tools/ab_pair.sh).   python tools/ab_pair.py [N C H K]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib, ops
L = _lib.lib(); dev = torch.device("cuda:0")
N, C, H, K = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (128, 96, 56, 51)
x = torch.randn(N, C, H, H, device=dev).bfloat16(); d1 = torch.randn_like(x); d2 = torch.randn_like(x)
dt = _lib.SLAK_BF16; st = torch.cuda.current_stream(dev).cuda_stream
nb = int(L.slak_dwconv2d_pair_filter_workspace_bytes(dt, N, C, H, H, K))
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
dwv = torch.empty(C, 1, K, 5, device=dev); dws = torch.empty(C, 1, 5, 5, device=dev); dwh = torch.empty(C, 1, 5, K, device=dev)
nb2 = int(L.slak_dwconv2d_workspace_bytes(2, N, C, H, H, 5, K, dt)); ws2 = torch.empty(nb2, dtype=torch.uint8, device=dev)
def pair(): _lib.check(L.slak_dwconv2d_pair_backward_filter(d1.data_ptr(), d2.data_ptr(), x.data_ptr(), dwv.data_ptr(), dws.data_ptr(), dt, N, C, H, H, K, ws.data_ptr(), nb, st))
def horiz(): _lib.check(L.slak_dwconv2d_backward_filter(d1.data_ptr(), dt, x.data_ptr(), dt, dwh.data_ptr(), N, C, H, H, 5, K, ws2.data_ptr(), nb2, st))
wa = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
for _ in range(200): wa @ wa
torch.cuda.synchronize()
def t(fn, reps=100):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True)
    a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3
print("%-60s pair %7.1f us   5xK %7.1f us" % (" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("SLAK_")), t(pair), t(horiz)))
