"""Dev: a few launches of slak_linear_gemm (both epilogues) at the SLaK-T bs-128 shapes of stages 2-4, for rocprofv3 --pmc passes (tools/pmc_linear_gemm.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import _lib
dev = torch.device("cuda:0"); L = _lib.lib(); st = torch.cuda.current_stream().cuda_stream
for C, HW in ((192, 28), (384, 14), (768, 7)):
    M = 128 * HW * HW
    t = torch.randn(M, C, device=dev).bfloat16(); w1 = (torch.randn(4 * C, C, device=dev) * 0.05).bfloat16(); b1 = torch.randn(4 * C, device=dev).bfloat16()
    dz = torch.randn(M, C, device=dev).bfloat16(); w2t = (torch.randn(4 * C, C, device=dev) * 0.05).bfloat16()
    y1 = torch.empty(M, 4 * C, device=dev, dtype=torch.bfloat16); a = torch.empty_like(y1); dy1 = torch.empty_like(y1); db = torch.empty(4 * C, device=dev)
    nb = L.slak_linear_gemm_workspace_bytes(M, 4 * C, C, 2); ws = torch.empty(max(nb, 16), device=dev, dtype=torch.uint8)
    for _ in range(4):
        _lib.check(L.slak_linear_gemm(t.data_ptr(), w1.data_ptr(), b1.data_ptr(), y1.data_ptr(), a.data_ptr(), None, None, M, 4 * C, C, 1, None, 0, st))
        _lib.check(L.slak_linear_gemm(dz.data_ptr(), w2t.data_ptr(), None, dy1.data_ptr(), None, y1.data_ptr(), db.data_ptr(), M, 4 * C, C, 2, ws.data_ptr(), nb, st))
torch.cuda.synchronize()
