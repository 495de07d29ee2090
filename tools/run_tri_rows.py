"""Dev: 30 launches of the one-launch three-branch weight gradient (and of the pair + 5 x K launches) at one shape, for rocprofv3 --kernel-trace.
python tools/run_tri_rows.py [N C H K]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); dt = _lib.SLAK_BF16
N, C, H, K = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (128, 96, 56, 51)
st = torch.cuda.current_stream(dev).cuda_stream
x = torch.randn(N, C, H, H, device=dev).bfloat16(); dys = [torch.randn_like(x) for _ in range(3)]
dws = [torch.empty(C, 1, kh, kw, device=dev) for kh, kw in ((K, 5), (5, K), (5, 5))]
nb = int(L.slak_dwconv2d_tri_filter_workspace_bytes(dt, N, C, H, H, K)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
nbp = int(L.slak_dwconv2d_pair_filter_workspace_bytes(dt, N, C, H, H, K)); wsp = torch.empty(max(nbp, 16), dtype=torch.uint8, device=dev)
nb2 = int(L.slak_dwconv2d_workspace_bytes(2, N, C, H, H, 5, K, dt)); ws2 = torch.empty(max(nb2, 16), dtype=torch.uint8, device=dev)
wa = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
for _ in range(100): wa @ wa
for _ in range(30):
    if nb: _lib.check(L.slak_dwconv2d_tri_backward_filter(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), dt, N, C, H, H, K, ws.data_ptr(), nb, st))
    if os.environ.get("WITH_PAIR"):
        _lib.check(L.slak_dwconv2d_pair_backward_filter(dys[0].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[2].data_ptr(), dt, N, C, H, H, K, wsp.data_ptr(), nbp, st))
        _lib.check(L.slak_dwconv2d_backward_filter(dys[1].data_ptr(), dt, x.data_ptr(), dt, dws[1].data_ptr(), N, C, H, H, 5, K, ws2.data_ptr(), nb2, st))
torch.cuda.synchronize()
