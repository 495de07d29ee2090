"""dev: the 49x5 weight gradient on the 28x28 stage (192 channels, N = 128) -- SLAK_MFMA_VWAVE=0 for the transposing kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops
dev = torch.device("cuda:0")
N, C, H, W = 128, 192, 28, 28
KK = 49
if os.environ.get("WG_SHAPE"):                        # e.g. WG_SHAPE=96,56,51 for the 56x56 stage
    C, H, KK = (int(v) for v in os.environ["WG_SHAPE"].split(",")); W = H
x = torch.randn(N, C, H, W, device=dev).bfloat16(); dy = torch.randn_like(x)
for kh, kw in ((KK, 5), (5, KK), (5, 5)):
    w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
    for _ in range(5): ops.dwconv2d_backward_filter(dy, x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): ops.dwconv2d_backward_filter(dy, x, w)
    e1.record(); torch.cuda.synchronize()
    print("vwave=%s wgs=%s  %dx%d wgrad %dx%d: %.1f us" % (os.environ.get("SLAK_MFMA_VWAVE", "1"), os.environ.get("SLAK_VWAVE_WGS", "2"), kh, kw, H, W, e0.elapsed_time(e1) / 50 * 1e3))
from slak_amd import _lib, block_ops
L = _lib.lib(); st = torch.cuda.current_stream(dev).cuda_stream
K = KK
dys = torch.randn_like(x); dwv = torch.empty(C, 1, K, 5, device=dev); dws = torch.empty(C, 1, 5, 5, device=dev)
nb = int(L.slak_dwconv2d_pair_filter_workspace_bytes(_lib.SLAK_BF16, N, C, H, W, K))
if nb:
    wsb, nbb = block_ops._workspace(nb, dev)
    def pair(): _lib.check(L.slak_dwconv2d_pair_backward_filter(dy.data_ptr(), dys.data_ptr(), x.data_ptr(), dwv.data_ptr(), dws.data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, wsb.data_ptr(), nbb, st))
    for _ in range(5): pair()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): pair()
    e1.record(); torch.cuda.synchronize()
    print("pair %dx5 + 5x5 wgrad %dx%d: %.1f us" % (K, H, W, e0.elapsed_time(e1) / 50 * 1e3))
    w5 = torch.randn(C, 1, 5, 5, device=dev) * 0.02
    ref_s = ops.dwconv2d_backward_filter(dys, x, w5); ref_v = ops.dwconv2d_backward_filter(dy, x, torch.randn(C, 1, K, 5, device=dev))
    print("pair vs separate: dw_v max diff %.3g (max %.3g), dw_s max diff %.3g (max %.3g)" % ((dwv - ref_v).abs().max().item(), ref_v.abs().max().item(), (dws - ref_s).abs().max().item(), ref_s.abs().max().item()))
