"""Event-timed: the data-gradient launch + the weight-gradient launch of a block's three branch convs vs the one-launch backward (14 x 14 class)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib

def main():
    dev = torch.device("cuda:0")
    L = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    for (N, C, H, W, K) in [(128, 384, 14, 14, 47), (128, 512, 14, 14, 47), (64, 384, 24 // 2, 12, 61), (128, 768, 7, 7, 13), (64, 1024, 7, 7, 13)]:
        x = torch.randn(N, C, H, W, device=dev).bfloat16()
        dys = [torch.randn(N, C, H, W, device=dev).bfloat16() for _ in range(3)]
        ws = [torch.randn(C, 1, kh, kw, device=dev) * 0.05 for kh, kw in ((K, 5), (5, K), (5, 5))]
        dws = [torch.empty_like(w) for w in ws]
        dx = torch.empty_like(x)
        nb = int(L.slak_dwconv2d_tri_filter_workspace_bytes(_lib.SLAK_BF16, N, C, H, W, K))
        wsb = torch.empty(nb, dtype=torch.uint8, device=dev)
        burn = torch.randn(4096, 4096, device=dev)
        def two():
            L.slak_dwconv2d_tri_backward_data(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(),
                                              dx.data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, st)
            L.slak_dwconv2d_tri_backward_filter(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(),
                                                dws[2].data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, wsb.data_ptr(), nb, st)
        def one():
            rc = L.slak_dwconv2d_tri_backward(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(),
                                              dx.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, wsb.data_ptr(), nb, st)
            assert rc == 0, rc
        have_one = L.slak_dwconv2d_tri_backward_supported(_lib.SLAK_BF16, N, C, H, W, K) == 1
        for name, fn in (("two launches", two), ("one launch", one), ("two launches", two), ("one launch", one)):
            if fn is one and not have_one:
                continue
            for _ in range(3):
                fn()
            for _ in range(20):
                burn @ burn
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                fn()
            e1.record(); torch.cuda.synchronize()
            print("%s  %-14s %7.2f us" % ((N, C, H, W, K), name, e0.elapsed_time(e1) * 1000 / 50), flush=True)

main()
