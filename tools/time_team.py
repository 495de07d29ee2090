"""Dev: the three-branch launches (slak_dwconv2d_tri_forward / _backward_data) on the 56x56 / 28x28 class: quick parity vs the per-branch
kernels, then HIP-event timing.  Environment: SLAK_TEAM_TRI=0 (round-2 kernels), SLAK_TEAM_NB, SLAK_TEAM_DBG."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops, _lib
dev = torch.device("cuda:0"); L = _lib.lib()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
check = "--check" in sys.argv
shapes = [(128, 96, 56, 56, 51), (128, 192, 28, 28, 49), (64, 192, 48, 48, 59), (64, 384, 24, 24, 57), (64, 128, 56, 56, 51)]
shapes = shapes[:int(os.environ.get("TEAM_SHAPES", "5"))]
if os.environ.get("TEAM_SHAPE"):
    shapes = [tuple(int(v) for v in os.environ["TEAM_SHAPE"].split(","))]
if "--small" in sys.argv:
    shapes = [(5, 3, 56, 56, 51), (9, 2, 28, 28, 49), (2, 2, 48, 40, 31), (3, 2, 32, 32, 31), (6, 2, 24, 24, 13), (7, 2, 28, 20, 13), (1, 1, 56, 56, 51), (130, 2, 28, 28, 49)]
tag = "team=%s nb=%s dbg=%s dual=%s stag=%s" % (os.environ.get("SLAK_TEAM_TRI", "1"), os.environ.get("SLAK_TEAM_NB", "-"), os.environ.get("SLAK_TEAM_DBG", "0"), os.environ.get("SLAK_TEAM_DUAL", "1"), os.environ.get("SLAK_TEAM_STAGGER", "0"))
for (N, C, H, W, K) in shapes:
    torch.manual_seed(0)
    x = torch.randn(N, C, H, W, device=dev).bfloat16(); dys = [torch.randn_like(x) for _ in range(3)]
    ws = [torch.randn(C, 1, kh, kw, device=dev) * 0.05 for kh, kw in ((K, 5), (5, K), (5, 5))]
    ys = [torch.empty_like(x) for _ in range(3)]; dx = torch.empty_like(x)
    kind = L.slak_dwconv2d_tri_supported(_lib.SLAK_BF16, N, C, H, W, K)
    def tf(): _lib.check(L.slak_dwconv2d_tri_forward(x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, st))
    def td(): _lib.check(L.slak_dwconv2d_tri_backward_data(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), dx.data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, st))
    if check:
        for y in ys: y.fill_(float("nan"))
        dx.fill_(float("nan"))
        tf(); td(); torch.cuda.synchronize()
        bad = []
        for y, w, nm in zip(ys, ws, ("Kx5", "5xK", "5x5")):
            ref = ops.dwconv2d_forward(x, w)
            if not torch.equal(y, ref): bad.append("fwd %s maxdiff %.3e nan %d" % (nm, (y.float() - ref.float()).abs().nan_to_num(9e9).max().item(), int(torch.isnan(y).sum())))
        ref = sum(ops.dwconv2d_backward_data(d, w).float() for d, w in zip(dys, ws))
        e = (dx.float() - ref).abs().nan_to_num(9e9).max().item() / max(1.0, ref.abs().max().item())
        if e > 2e-2: bad.append("dgrad rel err %.3e nan %d" % (e, int(torch.isnan(dx).sum())))
        print(tag, (N, C, H, W, K), "kind", kind, "OK" if not bad else "BAD: " + "; ".join(bad), flush=True)
        continue
    by = 3 * 2 * x.numel() * 2
    for what, fn in (("tri fwd", tf), ("tri dgrad", td)):
        try:
            fn()
        except Exception as e:
            print(tag, (N, C, H, W, K), what, "not available:", str(e)[-60:]); continue
        for _ in range(5): fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        print(tag, (N, C, H, W, K), "kind", kind, what, "%.1f us  %.2f TB/s per-op (frac %.3f)" % (best, by / best / 1e6, by / best / 1e6 / 8.0), flush=True)
