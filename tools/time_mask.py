"""tools/time_mask.py -- slak_mask_prune_and_grow on the SLaK-T mask set, each call from the same state (run under rocprofv3 for per-kernel times)."""
import ctypes, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slak_amd import _lib
from slak_amd.slak_model import slak_mask_set_shapes
dev = torch.device("cuda:0"); L = _lib.lib(); st = torch.cuda.current_stream(dev).cuda_stream
shapes = slak_mask_set_shapes("tiny")
g = torch.Generator(device=dev).manual_seed(99)
ws = [torch.randn(s, device=dev, generator=g) * 0.02 for s in shapes]
ms = [(torch.rand(s, device=dev, generator=g) < 0.6).float() for s in shapes]
for w, m in zip(ws, ms): w.mul_(m)
gs = [torch.randn(s, device=dev, generator=g) for s in shapes]
if os.environ.get("UNREACHABLE", "0") == "1":               # zero gradients on the taps a small plane never reaches
    for s_, g_ in zip(shapes, gs):
        if len(s_) == 4 and s_[2] > 13: g_[:, :, : s_[2] // 2 - 6].zero_(); g_[:, :, s_[2] // 2 + 7:].zero_()
segs = (_lib.MaskSegment * len(shapes))()
for i in range(len(shapes)):
    segs[i].weight, segs[i].mask, segs[i].grad, segs[i].momentum, segs[i].numel = ws[i].data_ptr(), ms[i].data_ptr(), gs[i].data_ptr(), None, ws[i].numel()
plan = ctypes.c_void_p(); _lib.check(L.slak_mask_plan_create(segs, len(shapes), ctypes.byref(plan)))
ws0 = [w.clone() for w in ws]; ms0 = [m.clone() for m in ms]
for mode in ("fresh", "repeat"):
    ts = []
    for r in range(8):
        if mode == "fresh" or r == 0:
            torch._foreach_copy_(ws, ws0); torch._foreach_copy_(ms, ms0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); _lib.check(L.slak_mask_prune_and_grow(plan, 0.3, st)); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(mode, " ".join("%.3f" % t for t in ts), "ms; median %.3f" % float(np.median(ts[2:])))
