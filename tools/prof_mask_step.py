"""Dev: host-side profile of Masking.step() (cProfile) on the bench model."""
import sys, os, types, cProfile, pstats, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
import slak_amd.slak_model as M
from slak_amd.sparse_core import CosineDecay, Masking
dev = torch.device("cuda:0")
M.Block.fused_tail = True; M.ReparamLargeKernelConv.fused_bn = True; M.LayerNorm.fused_cf = True
model = M.SLaK_tiny(kernel_size=[51, 49, 47, 13, 5], Decom=True, bn=True, drop_path_rate=0.1, lowp_dwconv=True).to(dev)
opt = torch.optim.AdamW(model.parameters(), lr=4e-3, fused=True)
margs = types.SimpleNamespace(device=str(dev), fix=False, update_frequency=2000, only_L=False, sparse_init="uniform", sparsity=0.4, distributed=False)
with contextlib.redirect_stdout(io.StringIO()):
    mask = Masking(opt, None, CosineDecay(0.3, 1000), prune_rate=0.3, prune_mode="magnitude", growth_mode="gradient", redistribution_mode="none", args=margs)
    mask.add_module(model)
x = torch.randn(16, 3, 224, 224, device=dev); y = torch.randint(0, 1000, (16,), device=dev)
for i in range(4):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = nn.functional.cross_entropy(model(x), y)
    loss.backward()
    torch.cuda.synchronize()
    if i == 3:
        pr = cProfile.Profile(); pr.enable()
    mask.step()
    if i == 3:
        pr.disable()
    opt.zero_grad(set_to_none=True)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
