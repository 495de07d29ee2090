"""Dev: which kernel family every (config, stage, branch, op) of the BASELINE configurations lands on (slak_debug_last_kernel)."""
import sys, os, ctypes, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
CONFIGS = {"cfg1_slak_t_224": (128, [(96, 56, 51), (192, 28, 49), (384, 14, 47), (768, 7, 13)]),
           "cfg3_slak_b_224": (64, [(128, 56, 51), (256, 28, 49), (512, 14, 47), (1024, 7, 13)]),
           "cfg4_slak_t_384": (64, [(96, 96, 61), (192, 48, 59), (384, 24, 57), (768, 12, 13)])}
def last(): return L.slak_debug_last_kernel().decode()
out = {}
for cfg, (N, stages) in CONFIGS.items():
    for si, (C, HW, K) in enumerate(stages):
        x = torch.randn(N, C, HW, HW, device=dev).bfloat16(); y = torch.empty_like(x)
        dt = _lib.SLAK_BF16
        for kh, kw in ((K, 5), (5, K), (5, 5)):
            w = torch.randn(C, 1, kh, kw, device=dev) * 0.02; dw = torch.empty_like(w)
            dims = (N, C, HW, HW, kh, kw)
            nb = max(int(L.slak_dwconv2d_workspace_bytes(op, *dims, dt)) for op in (0, 1, 2)); ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            key = "%s/s%d/%dx%d" % (cfg, si + 1, kh, kw)
            _lib.check(L.slak_dwconv2d_forward(x.data_ptr(), dt, w.data_ptr(), 0, y.data_ptr(), dt, *dims, ws.data_ptr(), ws.numel(), st)); out[key + "/fwd"] = last()
            _lib.check(L.slak_dwconv2d_backward_data(x.data_ptr(), dt, w.data_ptr(), 0, y.data_ptr(), dt, *dims, ws.data_ptr(), ws.numel(), st)); out[key + "/bwd_data"] = last()
            rc = L.slak_dwconv2d_backward_data_accumulate(x.data_ptr(), dt, w.data_ptr(), 0, y.data_ptr(), dt, *dims, ws.data_ptr(), ws.numel(), st); out[key + "/bwd_data_acc"] = last() if rc == 0 else "unsupported"
            _lib.check(L.slak_dwconv2d_backward_filter(x.data_ptr(), dt, x.data_ptr(), dt, dw.data_ptr(), *dims, ws.data_ptr(), ws.numel(), st)); out[key + "/bwd_filter"] = last()
        key = "%s/s%d/tri" % (cfg, si + 1)
        out[key + "/use_fwd"] = int(L.slak_dwconv2d_tri_supported_op(dt, N, C, HW, HW, K, 0)); out[key + "/use_bwd_data"] = int(L.slak_dwconv2d_tri_supported_op(dt, N, C, HW, HW, K, 1))
        wts = [torch.randn(C, 1, kh, kw, device=dev) * 0.02 for kh, kw in ((K, 5), (5, K), (5, 5))]; ys = [torch.empty_like(x) for _ in range(3)]
        if out[key + "/use_fwd"]:
            _lib.check(L.slak_dwconv2d_tri_forward(x.data_ptr(), wts[0].data_ptr(), wts[1].data_ptr(), wts[2].data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), dt, N, C, HW, HW, K, st)); out[key + "/fwd"] = last()
        if out[key + "/use_bwd_data"]:
            _lib.check(L.slak_dwconv2d_tri_backward_data(x.data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), wts[0].data_ptr(), wts[1].data_ptr(), wts[2].data_ptr(), ys[2].data_ptr(), dt, N, C, HW, HW, K, st)); out[key + "/bwd_data"] = last()
        nb = int(L.slak_dwconv2d_tri_filter_workspace_bytes(dt, N, C, HW, HW, K))
        if nb:
            ws = torch.empty(nb, dtype=torch.uint8, device=dev); dws = [torch.empty_like(w) for w in wts]
            _lib.check(L.slak_dwconv2d_tri_backward_filter(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), dt, N, C, HW, HW, K, ws.data_ptr(), nb, st)); out[key + "/bwd_filter"] = last()
        out[key + "/use_bwd"] = int(L.slak_dwconv2d_tri_backward_supported(dt, N, C, HW, HW, K))     # data gradient + the three weight gradients in one launch
        if out[key + "/use_bwd"]:
            _lib.check(L.slak_dwconv2d_tri_backward(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), wts[0].data_ptr(), wts[1].data_ptr(), wts[2].data_ptr(), ys[2].data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dws[2].data_ptr(), dt, N, C, HW, HW, K, ws.data_ptr(), nb, st)); out[key + "/bwd"] = last()
        nb = int(L.slak_dwconv2d_pair_filter_workspace_bytes(dt, N, C, HW, HW, K))
        if nb:
            ws = torch.empty(nb, dtype=torch.uint8, device=dev); dws = [torch.empty_like(wts[0]), torch.empty_like(wts[2])]
            rc = L.slak_dwconv2d_pair_backward_filter(x.data_ptr(), x.data_ptr(), x.data_ptr(), dws[0].data_ptr(), dws[1].data_ptr(), dt, N, C, HW, HW, K, ws.data_ptr(), nb, st)
            out["%s/s%d/pair/bwd_filter" % (cfg, si + 1)] = last() if rc == 0 else "unsupported"
torch.cuda.synchronize()
print(json.dumps(out, indent=0))
