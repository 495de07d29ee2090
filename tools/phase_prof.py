"""Dev: per-phase cycle counts of workgroup 0 of the DMA conv kernel (s_memtime)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
L.slak_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
names = ["wait_dma", "B1", "bread+mfma", "epilogue", "B2", "issue/store"]
for (C, H, kh, kw) in ((96, 56, 5, 51), (96, 56, 51, 5), (192, 28, 5, 49), (192, 28, 49, 5)):
    x = torch.randn(128, C, H, H, device=dev).bfloat16(); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
    for _ in range(3): ops.dwconv2d_forward(x, w)
    dbg = torch.zeros(32, dtype=torch.int64, device=dev)
    L.slak_debug_set_phase_buffer(dbg.data_ptr())
    ops.dwconv2d_forward(x, w); torch.cuda.synchronize()
    L.slak_debug_set_phase_buffer(None)
    d = dbg.cpu().view(4, 8)
    print("C%d %dx%d k%dx%d iters=%d  (cycles per iteration, per wave)" % (C, H, H, kh, kw, int(d[0, 6])))
    for wv in range(4):
        it = max(1, int(d[wv, 6]))
        print("   wave %d: " % wv + "  ".join("%s %6.0f" % (n, d[wv, k].item() / it) for k, n in enumerate(names)) + "   total %6.0f" % (d[wv, :6].sum().item() / it))
print("wgrad phases (cycles, wave 0 of workgroup 0): setup | main loop | diag epilogue | final")
for N in (6, 128):
  for (C, H, kh, kw) in ((96, 56, 5, 51), (96, 56, 51, 5), (192, 28, 5, 49)):
    x = torch.randn(N, C, H, H, device=dev).bfloat16(); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
    for _ in range(3): ops.dwconv2d_backward_filter(x, x, w)
    dbg = torch.zeros(32, dtype=torch.int64, device=dev)
    L.slak_debug_set_phase_buffer(dbg.data_ptr())
    ops.dwconv2d_backward_filter(x, x, w); torch.cuda.synchronize()
    L.slak_debug_set_phase_buffer(None)
    print("  N=%d C%d %dx%d k%dx%d: " % (N, C, H, H, kh, kw), dbg[:5].tolist())
