"""Dev: per-workgroup timeline of the team kernel (SLAK_TEAM_DBG=16 in a library built with SLAK_BUILD_DEFS=-DSLAK_TEAM_DEV): which workgroups
share a CU, and how their phases line up.  --dgrad: the data-gradient launch instead of the forward one (SLAK_STREAM_TRI=0 keeps the forward on the team kernel)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
L.slak_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
N, C, H, W, K = 128, 96, 56, 56, 51
x = torch.randn(N, C, H, W, device=dev).bfloat16()
ws = [torch.randn(C, 1, kh, kw, device=dev) * 0.05 for kh, kw in ((K, 5), (5, K), (5, 5))]
ys = [torch.empty_like(x) for _ in range(3)]
dys = [torch.randn_like(x) for _ in range(3)]
def td(): _lib.check(L.slak_dwconv2d_tri_backward_data(dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, st))
def tf0(): _lib.check(L.slak_dwconv2d_tri_forward(x.data_ptr(), ws[0].data_ptr(), ws[1].data_ptr(), ws[2].data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), _lib.SLAK_BF16, N, C, H, W, K, st))
tf = td if "--dgrad" in sys.argv else tf0
for _ in range(3): tf()
buf = torch.zeros(1024 * 64, dtype=torch.int64, device=dev)
L.slak_debug_set_phase_buffer(buf.data_ptr())
tf(); torch.cuda.synchronize()
L.slak_debug_set_phase_buffer(None)
b = buf.cpu().view(1024, 64)
rows = [(i, b[i]) for i in range(1024) if b[i, 2] != 0]
print("workgroups that ran:", len(rows))
t0 = min(int(r[2]) for _, r in rows)
cus = {}
for i, r in rows:
    hw = int(r[0]); xcc = int(r[1]) & 15
    key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)      # xcc, se, sh, cu
    cus.setdefault(key, []).append((i, hw & 15, (hw >> 4) & 3, r))
print("distinct CUs:", len(cus), "teams per CU histogram:", {k: sum(1 for v in cus.values() if len(v) == k) for k in (1, 2, 3, 4)})
for key in list(cus)[:6]:
    print("CU", key)
    for (i, slot, simd, r) in cus[key]:
        ts = [int(r[k]) - t0 for k in range(2, 20)]
        print("   wg %4d slot %d simd %d start %6d  A/B stamps (10 ns): %s  end %d" % (i, slot, simd, ts[0], " ".join(str(t) for t in ts[1:15]), int(r[63]) - t0))
ends = [int(r[63]) - t0 for _, r in rows]; starts = [int(r[2]) - t0 for _, r in rows]
print("start spread %d..%d, end spread %d..%d (10 ns units)" % (min(starts), max(starts), min(ends), max(ends)))
import statistics
ca = [int(r[4]) - int(r[3]) for _, r in rows]; io = [int(r[5]) - int(r[4]) for _, r in rows]
print("first compute phase median %d, first IO phase median %d (x10 ns)" % (statistics.median(ca), statistics.median(io)))
ca = [int(r[24]) - int(r[23]) for _, r in rows]; io = [int(r[25]) - int(r[24]) for _, r in rows]
print("11th compute phase median %d, IO phase median %d (x10 ns)" % (statistics.median(ca), statistics.median(io)))

clk = [(int(r[61]) - int(r[60])) / max(1, (int(r[63]) - int(r[2]))) * 100.0 for _, r in rows]      # cycles per 10 ns -> MHz
print("effective shader clock over the kernel: median %.0f MHz (min %.0f, max %.0f)" % (statistics.median(clk), min(clk), max(clk)))
body = [int(r[58]) - int(r[57]) for _, r in rows]; wt = [int(r[59]) - int(r[58]) for _, r in rows]
print("compute phase 10 of wave 0: tiles %d cycles (median), DMA wait %d cycles" % (statistics.median(body), statistics.median(wt)))
t0c = [int(r[56]) - int(r[55]) for _, r in rows if int(r[56])]
if t0c: print("first tile (20 MFMAs) of compute phase 10, wave 0: %d cycles (median), prologue-to-first-tile %d" % (statistics.median(t0c), statistics.median([int(r[55]) - int(r[57]) for _, r in rows if int(r[56])])))
io = [(int(r[53]) - int(r[52]), int(r[54]) - int(r[53]), int(r[51]) - int(r[54])) for _, r in rows if int(r[51])]
if io: print("IO phase 10, wave 0: DMA issue %d, copy-out %d, transposes %d cycles (medians)" % tuple(statistics.median(x[k] for x in io) for k in range(3)))

