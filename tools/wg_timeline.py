"""Dev: per-workgroup real-time stamps (100 MHz) of the DMA conv kernel: entry / loop start / loop end."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops, _lib
L = _lib.lib(); dev = torch.device("cuda:0")
L.slak_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
for (C, H, kh, kw) in ((96, 56, 5, 51), (96, 56, 51, 5), (192, 28, 5, 49)):
    x = torch.randn(128, C, H, H, device=dev).bfloat16(); w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
    for _ in range(3): ops.dwconv2d_forward(x, w)
    dbg = torch.zeros(64 + 8 * 4096, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    L.slak_debug_set_phase_buffer(dbg.data_ptr())
    e0.record(); ops.dwconv2d_forward(x, w); e1.record(); torch.cuda.synchronize()
    L.slak_debug_set_phase_buffer(None)
    d = dbg[64:].cpu().view(-1, 8); d = d[d[:, 0] > 0]
    t0 = d[:, 0].min().item()
    ent, ls, le = (d[:, 0] - t0).float() / 100, (d[:, 1] - t0).float() / 100, (d[:, 2] - t0).float() / 100
    print("C%d %dx%d k%dx%d: %d WGs, events %.1f us | entry min/med/max %.1f %.1f %.1f | loop start %.1f %.1f %.1f | loop end %.1f %.1f %.1f (us)" % (
        C, H, H, kh, kw, d.shape[0], e0.elapsed_time(e1) * 1e3, ent.min(), ent.median(), ent.max(), ls.min(), ls.median(), ls.max(), le.min(), le.median(), le.max()))
    full = dbg[64:].cpu().view(-1, 8)[:d.shape[0]]
    st = [((full[:, k] - full[:, 0]).float() / 100).median().item() for k in (3, 4, 5, 6, 1)]
    print('   prologue medians (us since entry): descr+maps %.2f | dma issued+zero fill %.2f | windows written %.2f | frags built %.2f | loop start %.2f' % tuple(st))
    import numpy as np
    clk = (full[:, 7].float() / ((full[:, 2] - full[:, 1]).float() * 10)).numpy()   # cycles per ns
    print('   shader clock during the loop (GHz): min %.2f med %.2f max %.2f' % (clk.min(), np.median(clk), clk.max()))
    bid = np.arange(d.shape[0]); dur = ((full[:, 2] - full[:, 1]).float() / 100).numpy()
    print("   loop duration by XCD (bid%8):", " ".join("%.1f" % dur[bid % 8 == k].mean() for k in range(8)))
    sl = bid // C
    print("   loop duration by slice:", " ".join("%.1f" % dur[sl == k].mean() for k in range(sl.max() + 1)))
    print("   by (bid//8)%32 (CU within XCD?):", " ".join("%.1f" % dur[(bid // 8) % 32 == k].mean() for k in range(32)))
    print("   percentiles 5/25/50/75/95: ", " ".join("%.1f" % np.percentile(dur, q) for q in (5, 25, 50, 75, 95)))
