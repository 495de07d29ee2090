"""dev: the stage-1 pointwise GEMMs through torch with the TunableOp solutions bench.py uses vs slak_linear_nt"""
import sys, os, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d = tempfile.mkdtemp()
shutil.copy(os.path.join(ROOT, "slak_amd", "tuning", "tunableop_gfx950.csv"), os.path.join(d, "tunableop_results0.csv"))
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="0", PYTORCH_TUNABLEOP_RECORD_UNTUNED="0", PYTORCH_TUNABLEOP_FILENAME=os.path.join(d, "tunableop_results.csv"))
import torch, torch.nn.functional as F
from slak_amd import block_ops
block_ops.use_skinny_linear = True
dev = torch.device("cuda:0")
def ev(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
C, HW = 96, 56
M = 128 * HW * HW
t = torch.randn(M, C, device=dev).bfloat16(); w1 = torch.randn(4 * C, C, device=dev).bfloat16(); b1 = torch.randn(4 * C, device=dev).bfloat16()
w2 = torch.randn(C, 4 * C, device=dev).bfloat16(); b2 = torch.randn(C, device=dev).bfloat16()
a = torch.randn(M, 4 * C, device=dev).bfloat16(); dz = torch.randn(M, C, device=dev).bfloat16()
w2t = w2.t().contiguous(); w1t = w1.t().contiguous()
for name, fn, byt in (("fwd1 tuned", lambda: F.linear(t, w1, b1), 2 * M * 5 * C), ("gelu", lambda: F.gelu(a), 2 * M * 8 * C),
                      ("fwd1+gelu skinny", lambda: block_ops.linear_nt(t, w1, b1, gelu=True), 2 * M * 9 * C),
                      ("fwd2 tuned", lambda: F.linear(a, w2, b2), 2 * M * 5 * C), ("fwd2 skinny", lambda: block_ops.linear_nt(a, w2, b2), 2 * M * 5 * C),
                      ("dact tuned", lambda: torch.mm(dz, w2), 2 * M * 5 * C), ("dact skinny", lambda: block_ops.linear_nt(dz, w2t), 2 * M * 5 * C),
                      ("dt tuned", lambda: torch.mm(a, w1), 2 * M * 5 * C), ("dt skinny", lambda: block_ops.linear_nt(a, w1t), 2 * M * 5 * C)):
    us = ev(fn)
    print("%-20s %7.1f us  %.2f TB/s" % (name, us, byt / us / 1e6))
print("tunable results in use:", len(torch.cuda.tunable.get_results()))
