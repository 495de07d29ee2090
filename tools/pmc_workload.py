"""Workload for the rocprofv3 PMC passes: every SLaK-T dw-conv kernel shape (bf16, N=128), 3 launches each, then the launches as the
model runs them that the per-branch list does not contain: the accumulating data gradient (stages 1-2) and the three-branch launches
(stages 3-4: forward, data gradient, weight gradient), the K x 5 + 5 x 5 weight-gradient launch (stages 1-2).
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_fetch -o pmc -- python tools/pmc_workload.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_write -o pmc -- python tools/pmc_workload.py
then tools/pmc_traffic.py turns the two databases into profiles/pmc_traffic.json."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (C, H, K) in ((96, 56, 51), (192, 28, 49), (384, 14, 47), (768, 7, 13)):
    x = torch.randn(128, C, H, H, device=dev).bfloat16(); dy = torch.randn_like(x)
    for (kh, kw) in ((K, 5), (5, K), (5, 5)):
        w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
        for _ in range(3):
            ops.dwconv2d_forward(x, w); ops.dwconv2d_backward_data(dy, w); ops.dwconv2d_backward_filter(dy, x, w)
        torch.cuda.synchronize()

# ---- the launches of a training step that are not in the per-branch list above (tools/pmc_traffic.py reads them in this order) ----
from slak_amd import block_ops
for (C, H, K) in ((96, 56, 51), (192, 28, 49)):
    dy = torch.randn(128, C, H, H, device=dev).bfloat16(); dx = torch.randn_like(dy)
    for (kh, kw) in ((5, K), (5, 5)):
        w = torch.randn(C, 1, kh, kw, device=dev) * 0.02
        for _ in range(3):
            ops.dwconv2d_backward_data_accumulate(dy, w, dx)
        torch.cuda.synchronize()
for (C, H, K) in ((384, 14, 47), (768, 7, 13)):
    x = torch.randn(128, C, H, H, device=dev).bfloat16().requires_grad_(True)
    ws = [(torch.randn(C, 1, kh, kw, device=dev) * 0.02).requires_grad_(True) for kh, kw in ((K, 5), (5, K), (5, 5))]
    dys = [torch.randn(128, C, H, H, device=dev).bfloat16() for _ in range(3)]
    for _ in range(3):
        ys = block_ops.tri_dwconv(x, *ws)                       # one launch
        torch.autograd.backward(ys, dys)                        # one data-gradient launch, one weight-gradient launch
    torch.cuda.synchronize()
# ---- the K x 5 and the 5 x 5 weight gradient of a block in one launch (stages 1-2) ----
from slak_amd import _lib
L = _lib.lib(); st = torch.cuda.current_stream(dev).cuda_stream
for (C, H, K) in ((96, 56, 51), (192, 28, 49)):
    x = torch.randn(128, C, H, H, device=dev).bfloat16(); dyv = torch.randn_like(x); dys_ = torch.randn_like(x)
    dwv = torch.empty(C, 1, K, 5, device=dev); dws = torch.empty(C, 1, 5, 5, device=dev)
    nb = int(L.slak_dwconv2d_pair_filter_workspace_bytes(_lib.SLAK_BF16, 128, C, H, H, K))
    wsb, nbb = block_ops._workspace(nb, dev)
    for _ in range(3):
        _lib.check(L.slak_dwconv2d_pair_backward_filter(dyv.data_ptr(), dys_.data_ptr(), x.data_ptr(), dwv.data_ptr(), dws.data_ptr(), _lib.SLAK_BF16, 128, C, H, H, K, wsb.data_ptr(), nbb, st))
    torch.cuda.synchronize()
