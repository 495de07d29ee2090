"""Dev: the one-launch three-branch weight gradient on 2 x 2-tile planes against the two-launch path (pair + single), the C oracle on small
shapes, and event timing at the bench shapes.   python tools/check_tri_rows.py"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from slak_amd import _lib, ops
import oracle
L = _lib.lib(); dev = torch.device("cuda:0"); dt = _lib.SLAK_BF16
def st(): return torch.cuda.current_stream(dev).cuda_stream
def tri(dys, x, K):
    N, C, H, W = x.shape
    nb = int(L.slak_dwconv2d_tri_filter_workspace_bytes(dt, N, C, H, W, K)); assert nb > 0, (N, C, H, W, K)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dws = [torch.full((C, 1, kh, kw), float("nan"), device=dev) for kh, kw in ((K, 5), (5, K), (5, 5))]
    _lib.check(L.slak_dwconv2d_tri_backward_filter(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(),
                                                   dt, N, C, H, W, K, ws.data_ptr(), nb, st()))
    return dws, L.slak_debug_last_kernel()
for (N, C, H, W, K) in [(1, 1, 56, 56, 51), (3, 2, 56, 56, 51), (5, 3, 56, 56, 51), (2, 2, 64, 64, 61), (3, 2, 48, 48, 59), (2, 3, 40, 48, 31), (2, 2, 56, 40, 13), (7, 2, 64, 56, 51), (129, 3, 56, 56, 51), (7, 5, 56, 56, 51), (3, 11, 48, 48, 59),
                        (9, 2, 28, 28, 49), (1, 1, 28, 28, 49), (11, 3, 28, 28, 49), (5, 2, 24, 24, 57), (7, 2, 28, 20, 13), (3, 2, 32, 32, 31), (130, 7, 28, 28, 49), (2, 300, 24, 24, 13)]:
    torch.manual_seed(N + K)
    x = torch.randn(N, C, H, W, device=dev).bfloat16(); dys = [torch.randn(N, C, H, W, device=dev).bfloat16() for _ in range(3)]
    dws, name = tri(dys, x, K)
    xr = x.float().cpu().numpy(); worst = 0.0
    for dw, dy, (kh, kw) in zip(dws, dys, ((K, 5), (5, K), (5, 5))):
        ref = oracle.dwconv2d_bwd_filter(dy.float().cpu().numpy(), xr, kh, kw)
        err = np.abs(dw.double().cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()); worst = max(worst, err)
    dws2, _ = tri(dys, x, K)
    print("%-26s %s  max rel err vs oracle %.2e  reproducible %s" % ((N, C, H, W, K), name.decode(), worst, all(torch.equal(a, b) for a, b in zip(dws, dws2))))
def t(fn, reps=50):
    for _ in range(10): fn()
    a, b = torch.cuda.Event(True), torch.cuda.Event(True); a.record()
    for _ in range(reps): fn()
    b.record(); b.synchronize(); return a.elapsed_time(b) / reps * 1e3
wa = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
for _ in range(200): wa @ wa
for (N, C, H, K) in [(128, 96, 56, 51), (64, 128, 56, 51), (64, 192, 48, 59), (128, 192, 28, 49), (64, 256, 28, 49), (64, 384, 24, 57)]:
    x = torch.randn(N, C, H, H, device=dev).bfloat16(); dys = [torch.randn_like(x) for _ in range(3)]
    nb = int(L.slak_dwconv2d_tri_filter_workspace_bytes(dt, N, C, H, H, K)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    dws = [torch.empty(C, 1, kh, kw, device=dev) for kh, kw in ((K, 5), (5, K), (5, 5))]
    f = lambda: _lib.check(L.slak_dwconv2d_tri_backward_filter(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), dt, N, C, H, H, K, ws.data_ptr(), nb, st()))
    nbp = int(L.slak_dwconv2d_pair_filter_workspace_bytes(dt, N, C, H, H, K)); wsp = torch.empty(max(nbp, 16), dtype=torch.uint8, device=dev)
    nb2 = int(L.slak_dwconv2d_workspace_bytes(2, N, C, H, H, 5, K, dt)); ws2 = torch.empty(max(nb2, 16), dtype=torch.uint8, device=dev)
    g = lambda: (_lib.check(L.slak_dwconv2d_pair_backward_filter(dys[0].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[2].data_ptr(), dt, N, C, H, H, K, wsp.data_ptr(), nbp, st())),
                 _lib.check(L.slak_dwconv2d_backward_filter(dys[1].data_ptr(), dt, x.data_ptr(), dt, dws[1].data_ptr(), N, C, H, H, 5, K, ws2.data_ptr(), nb2, st())))
    S = x.numel()
    tt, tg = t(f), t(g)
    print("%-22s one launch %7.1f us (%.3f of 8 TB/s on 3 x 2 S b)   pair + 5xK %7.1f us   env %s" % ((N, C, H, K), tt, 6 * S * 2 / tt / 1e6 / 8000, tg, os.environ.get("SLAK_TRIROWS_DBG")))
