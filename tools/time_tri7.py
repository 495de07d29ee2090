"""dev: time the three-branch launches on the 7x7 stage (768 channels, N = 128) -- SLAK_SMALL_QUAD=0 for the one-plane-per-tile kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = torch.cuda.current_stream(dev).cuda_stream
N, C, H, W, K = 128, 768, 7, 7, 13
if os.environ.get("TRI_SHAPE"):                      # e.g. TRI_SHAPE=384,14,47 for the 14x14 stage
    C, H, K = (int(v) for v in os.environ["TRI_SHAPE"].split(",")); W = H
x = torch.randn(N, C, H, W, device=dev).bfloat16(); dys = [torch.randn_like(x) for _ in range(3)]
ws = [torch.randn(C, 1, kh, kw, device=dev) * 0.02 for kh, kw in ((K, 5), (5, K), (5, 5))]
ys = [torch.empty_like(x) for _ in range(3)]; dx = torch.empty_like(x)
def tf(): _lib.check(L.slak_dwconv2d_tri_forward(x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, st))
def td(): _lib.check(L.slak_dwconv2d_tri_backward_data(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), dx.data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, st))
for what, fn in (("tri fwd %dx%d" % (H, W), tf), ("tri dgrad %dx%d" % (H, W), td)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print("quad=%s" % os.environ.get("SLAK_SMALL_QUAD", "1"), what, "%.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
dws = [torch.empty_like(w) for w in ws]
wsb = L.slak_dwconv2d_tri_filter_workspace_bytes(_lib.SLAK_BF16, N, C, H, W, K)
wsp = torch.empty(max(int(wsb), 16), dtype=torch.uint8, device=dev)
def tw(): _lib.check(L.slak_dwconv2d_tri_backward_filter(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, wsp.data_ptr(), int(wsb), st))
for _ in range(5): tw()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): tw()
e1.record(); torch.cuda.synchronize()
print("quad=%s" % os.environ.get("SLAK_SMALL_QUAD", "1"), "tri wgrad %dx%d" % (H, W), "%.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
