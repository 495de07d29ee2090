"""dev: slak_linear_gemm_{gelu,dgelu} vs the tuned library GEMM + elementwise kernels at the stage 2-4 shapes; correctness vs fp64"""
import sys, os, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d = tempfile.mkdtemp()
shutil.copy(os.path.join(ROOT, "slak_amd", "tuning", "tunableop_gfx950.csv"), os.path.join(d, "tunableop_results0.csv"))
os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="0", PYTORCH_TUNABLEOP_RECORD_UNTUNED="0", PYTORCH_TUNABLEOP_FILENAME=os.path.join(d, "tunableop_results.csv"))
import torch, torch.nn.functional as F
from slak_amd import block_ops, _lib
dev = torch.device("cuda:0"); L = _lib.lib()
def ev(fn, reps=20):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
for (M, C) in ((777, 192), (100352, 192), (25088, 384), (6272, 768)):
    N, K = 4 * C, C
    x = torch.randn(M, K, device=dev).bfloat16(); w1 = (torch.randn(N, K, device=dev) * 0.05).bfloat16(); b1 = torch.randn(N, device=dev).bfloat16()
    y, g = block_ops.linear_gemm_gelu(x, w1, b1)
    ref = (x.double() @ w1.double().t() + b1.double())
    e1 = ((y.double() - ref).abs() / (ref.abs() * 2.0 ** -8 + 1e-2)).max().item()
    gw = F.gelu(y.float()).bfloat16()
    e2 = (g.float() - gw.float()).abs().max().item(); ex = (g == gw).float().mean().item()
    dz = torch.randn(M, K, device=dev).bfloat16(); w2t = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    dy1, db1 = block_ops.linear_gemm_dgelu(dz, w2t, y)
    dact = dz.double() @ w2t.double().t()
    yy = y.double(); gp = 0.5 * (1 + torch.erf(yy / 2 ** 0.5)) + yy * torch.exp(-0.5 * yy * yy) / (2 * 3.141592653589793) ** 0.5
    rd = dact * gp
    e3 = ((dy1.double() - rd).abs() / (rd.abs() * 2.0 ** -8 + 1e-3)).max().item()
    e4 = ((db1.double() - dy1.double().sum(0)).abs().max() / dy1.double().sum(0).abs().max()).item()
    print("M %6d C %3d: y err/tol %.2f  gelu max|d| %.2e exact %.4f  dy1 err/tol %.2f  db1 rel %.1e" % (M, C, e1, e2, ex, e3, e4))
    if M < 1000: continue
    tl = ev(lambda: F.linear(x, w1, b1)); tg = ev(lambda: F.gelu(y)); tm = ev(lambda: block_ops.linear_gemm_gelu(x, w1, b1))
    w2 = w2t.t().contiguous()
    td = ev(lambda: torch.mm(dz, w2))
    dact16 = torch.mm(dz, w2); dy = torch.empty_like(dact16); db = torch.empty(N, device=dev)
    ws, nb = block_ops._workspace(L.slak_gelu_bwd_workspace_bytes(M, N), dev); st = torch.cuda.current_stream().cuda_stream
    tb = ev(lambda: L.slak_gelu_backward_bias(dact16.data_ptr(), y.data_ptr(), dy.data_ptr(), db.data_ptr(), M, N, ws.data_ptr(), nb, st))
    tm2 = ev(lambda: block_ops.linear_gemm_dgelu(dz, w2t, y))
    print("     fwd: library %.1f + gelu %.1f = %.1f us   fused %.1f us      bwd: library %.1f + gelu_bwd %.1f = %.1f us   fused %.1f us" % (tl, tg, tl + tg, tm, td, tb, td + tb, tm2))
