"""Event-timed stem backward pieces at SLaK-T's shape (N = 128, 224 px): slak_stem_wgrad against torch's per-image GEMM + sums,
slak_channel_sums_bf16 against torch's sum((0, 2)) on the gradients of the four stem / downsample convolutions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from slak_amd import block_ops, _lib
dev = torch.device("cuda:0")
L = _lib.lib()


def timed(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps


N = int(os.environ.get("BATCH", "128"))
for (Co, P, K) in [(96, 3136, 48), (128, 3136, 48)]:
    dy = torch.randn(N, Co, P, device=dev).bfloat16(); a = torch.randn(N, P, K, device=dev).bfloat16()
    nb = int(L.slak_stem_wgrad_workspace_bytes(N, Co, P, K)); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    dw = torch.empty(Co, K, device=dev); db = torch.empty(Co, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    t_new = timed(lambda: L.slak_stem_wgrad(dy.data_ptr(), a.data_ptr(), dw.data_ptr(), db.data_ptr(), N, Co, P, K, ws.data_ptr(), nb, st))
    t_old = timed(lambda: (torch.bmm(dy, a).sum(0, dtype=torch.float32), dy.sum((0, 2), dtype=torch.float32)))
    byts = (dy.numel() + a.numel()) * 2
    print("stem wgrad Co=%d: own %.1f us (%.0f GB/s)   torch bmm + sums %.1f us" % (Co, t_new, byts / t_new / 1e3, t_old), flush=True)
for (C, P) in [(96, 3136), (192, 784), (384, 196), (768, 49)]:
    dy = torch.randn(N, C, P, device=dev).bfloat16()
    t_new = timed(lambda: block_ops.channel_sums(dy))
    t_old = timed(lambda: dy.sum((0, 2), dtype=torch.float32))
    print("channel sums C=%d P=%d: own %.1f us (%.0f GB/s)   torch %.1f us" % (C, P, t_new, dy.numel() * 2 / t_new / 1e3, t_old), flush=True)
for Co in (96, 128):
    x = torch.randn(N, 3, 224, 224, device=dev); w = torch.randn(Co, 3, 4, 4, device=dev) * 0.1; b = torch.randn(Co, device=dev) * 0.1
    a = torch.empty(N, 3136, 48, device=dev, dtype=torch.bfloat16); y = torch.empty(N, Co, 56, 56, device=dev, dtype=torch.bfloat16)
    st = torch.cuda.current_stream(dev).cuda_stream
    t_new = timed(lambda: L.slak_stem_conv_forward(x.data_ptr(), w.data_ptr(), b.data_ptr(), a.data_ptr(), y.data_ptr(), N, 3, 224, 224, Co, st))

    def old():
        L.slak_stem_patchify(x.data_ptr(), a.data_ptr(), N, 3, 224, 224, st)
        wp = w.reshape(Co, 48).to(torch.bfloat16)
        return torch.baddbmm(b.to(torch.bfloat16).view(1, Co, 1).expand(N, Co, 3136), wp.unsqueeze(0).expand(N, Co, 48), a.transpose(1, 2))
    t_old = timed(old)
    byts = x.numel() * 4 + a.numel() * 2 + y.numel() * 2
    print("stem conv forward Co=%d: own %.1f us (%.0f GB/s)   patchify + library GEMM %.1f us" % (Co, t_new, byts / t_new / 1e3, t_old), flush=True)
