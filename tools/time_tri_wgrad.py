import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops, _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = torch.cuda.current_stream(dev).cuda_stream
for (N, C, H, W, K) in [(128, 384, 14, 14, 47), (128, 768, 7, 7, 13), (64, 768, 12, 12, 13)]:
    x = torch.randn(N, C, H, W, device=dev).bfloat16(); dys = [torch.randn_like(x) for _ in range(3)]
    ws = [torch.randn(C, 1, kh, kw, device=dev) * 0.02 for kh, kw in ((K, 5), (5, K), (5, 5))]
    dws = [torch.empty_like(w) for w in ws]
    nb = int(L.slak_dwconv2d_tri_filter_workspace_bytes(_lib.SLAK_BF16, N, C, H, W, K)); wsb = torch.empty(nb, dtype=torch.uint8, device=dev)
    def tw(): _lib.check(L.slak_dwconv2d_tri_backward_filter(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, wsb.data_ptr(), nb, st))
    def sw():
        for d, w in zip(dys, ws): ops.dwconv2d_backward_filter(d, x, w)
    for what, fn in (("tri wgrad", tw), ("3x wgrad", sw)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        byt = 3 * 2 * N * C * H * W * 2
        print((N, C, H, W, K), what, "%.1f us; per-op alg bytes %.0f MB -> frac %.3f" % (us, byt / 1e6, byt / us / 1e6 / 8))
