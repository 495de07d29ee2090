"""dev: the pointwise weight-gradient GEMMs of a bs-128 SLaK-T step: slak_linear_wgrad vs the split-K batched library GEMM bench.py used"""
import sys, os, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d = tempfile.mkdtemp()
shutil.copy(os.path.join(ROOT, "slak_amd", "tuning", "tunableop_gfx950.csv"), os.path.join(d, "tunableop_results0.csv"))
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="0", PYTORCH_TUNABLEOP_RECORD_UNTUNED="0", PYTORCH_TUNABLEOP_FILENAME=os.path.join(d, "tunableop_results.csv"))
import torch
from slak_amd import block_ops
dev = torch.device("cuda:0")
def ev(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for C, HW in ((96, 56), (192, 28), (384, 14), (768, 7)):
    M = 128 * HW * HW
    for N1, N2 in ((4 * C, C), (C, 4 * C)):
        dy = torch.randn(M, N1, device=dev).bfloat16(); x = torch.randn(M, N2, device=dev).bfloat16()
        S = max(1, M // 6272)
        lib = ev(lambda: torch.bmm(dy.view(S, M // S, -1).transpose(1, 2), x.view(S, M // S, -1)).sum(0, dtype=torch.float32))
        mine = ev(lambda: block_ops.linear_wgrad(dy, x))
        fl = 2.0 * M * N1 * N2; by = 2.0 * M * (N1 + N2)
        print("M %6d  %4d x %4d: library split-K %6.1f us   slak_linear_wgrad %6.1f us  (%.0f TFLOP/s, %.2f TB/s of operand bytes)" % (M, N1, N2, lib, mine, fl / mine / 1e6, by / mine / 1e6))
