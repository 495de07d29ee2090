"""Dev: effective shader clock inside the one-launch weight gradient (SLAK_TRIROWS_DBG=36[+2..]): s_memtime cycles / s_memrealtime 100 MHz ticks."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); dt = _lib.SLAK_BF16
N, C, H, K = 128, 96, 56, 51
st = torch.cuda.current_stream(dev).cuda_stream
x = torch.randn(N, C, H, H, device=dev).bfloat16(); dys = [torch.randn_like(x) for _ in range(3)]
dws = [torch.zeros(C, 1, kh, kw, device=dev) for kh, kw in ((K, 5), (5, K), (5, 5))]
nb = int(L.slak_dwconv2d_tri_filter_workspace_bytes(dt, N, C, H, H, K)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
for _ in range(20):
    _lib.check(L.slak_dwconv2d_tri_backward_filter(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), dt, N, C, H, H, K, ws.data_ptr(), nb, st))
torch.cuda.synchronize()
v = dws[2].flatten()[:16].cpu().tolist()
for b in range(8):
    print("wg %d: %.0f cycles in %.2f us -> %.3f GHz" % (b, v[2 * b], v[2 * b + 1] / 100.0, v[2 * b] / max(1.0, v[2 * b + 1]) / 10.0))
