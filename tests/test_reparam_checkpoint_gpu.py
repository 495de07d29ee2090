"""SURVEY.md 8f-4 on the GPU: inference re-parameterisation of the decomposed block, mask persistence, checkpoint round trip."""
import contextlib
import io
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
sys.path.insert(0, GOLDEN)


def _randomize_bn(mod, gen):
    for m in mod.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.SyncBatchNorm)):
            with torch.no_grad():
                m.weight.copy_(torch.rand(m.weight.shape, generator=gen) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=gen) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=gen) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=gen) + 0.5)


@pytest.mark.parametrize("C,K,HW", [(24, 47, 14), (8, 51, 56), (16, 13, 7), (12, 49, 28)])
def test_merge_kernel_decomposed_path(C, K, HW, gpu):
    """eval-mode output of LoRA1 + LoRA2 + small_conv (conv -> BN each, models/SLaK.py:92-95) == merged thin kernels + one bias.
    fp32: both sides run the fp32-exact kernels, difference is the re-association of the BN scale (1e-5 of the output range).
    bf16: the merged one-launch path against the fp32 result, BASELINE's 1e-2 bf16 tolerance."""
    import slak_amd.slak_model as M
    old = M.use_sync_bn
    M.use_sync_bn = False
    try:
        gen = torch.Generator().manual_seed(C + K)
        blk = M.ReparamLargeKernelConv(C, C, K, 1, C, 5, Decom=True, bn=True)
        _randomize_bn(blk, gen)
        blk = blk.to(gpu).eval()
        x = torch.randn(3, C, HW, HW, generator=gen).to(gpu)
        with torch.no_grad():
            ref = blk(x)
            keys_before = set(blk.state_dict().keys())
            blk.merge_kernel()
            got = blk(x)
            got16 = blk(x.to(torch.bfloat16))
        assert "LoRA1.conv.weight" in keys_before
        assert set(blk.state_dict().keys()) == {"LoRA1_reparam.weight", "LoRA2_reparam.weight", "small_conv_reparam.weight", "reparam_bias"}
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 1e-5 * scale
        assert got16.dtype == torch.bfloat16
        assert float((got16.float() - ref).abs().max()) <= 1e-2 * scale
    finally:
        M.use_sync_bn = old


def _args(device, update_frequency=3, sparsity=0.5):
    return types.SimpleNamespace(device=str(device), fix=False, update_frequency=update_frequency, only_L=False, sparse_init="uniform",
                                 sparsity=sparsity, distributed=False, output_dir=None, save_ckpt_num=3, save_ckpt_freq=1, resume='',
                                 auto_resume=True)


def _make(gpu, seed):
    from make_golden import TinyNet
    from slak_amd.optim_factory import MaskedAdamW
    from slak_amd.sparse_core import CosineDecay, Masking
    torch.manual_seed(seed)
    model = TinyNet()
    for p in model.parameters():
        p.data = torch.randn_like(p) * 0.05
    model = model.to(gpu)
    opt = MaskedAdamW(model.parameters(), lr=1e-2, weight_decay=0.05)
    with contextlib.redirect_stdout(io.StringIO()):
        mask = Masking(opt, None, CosineDecay(0.4, 20), prune_rate=0.4, prune_mode="magnitude", growth_mode="gradient",
                       redistribution_mode="none", args=_args(gpu))
        mask.add_module(model)
    return model, opt, mask


def test_checkpoint_resumes_with_identical_masks(tmp_path, gpu):
    """Run 3 steps (prune-and-grow closes the third), checkpoint, run 5 more; a FRESH model / optimizer / Masking / EMA restored from the checkpoint
    and fed the same gradients ends bit-identical: masks, weights, optimizer state, EMA.  `weight != 0` (--sparse_init resume) would
    not: regrown weights sit at exactly 0 under mask == 1."""
    from slak_amd import checkpoint
    from slak_amd.model_sema import ModelEma
    gen = torch.Generator().manual_seed(4)
    model, opt, mask = _make(gpu, 1)
    ema = ModelEma(model, decay=0.99)
    grads = [[(torch.randn(p.shape, generator=gen) * 0.1).to(gpu) for p in model.parameters()] for _ in range(8)]

    def run(model, opt, mask, ema, steps):
        for s in steps:
            for p, g in zip(model.parameters(), grads[s]):
                p.grad = g.clone()
            with contextlib.redirect_stdout(io.StringIO()):
                mask.step()
            opt.zero_grad(set_to_none=True)
            ema.update(model, mask)

    run(model, opt, mask, ema, range(3))
    lost = sum(int(((mask.masks[n] != 0) & (dict(model.named_parameters())[n] == 0)).sum()) for n in mask.masks)
    assert lost > 0                                            # the case `weight != 0` cannot represent
    args = _args(gpu)
    args.output_dir = str(tmp_path)
    checkpoint.save_model(args, 0, model, model, opt, None, model_ema=ema, mask=mask)
    run(model, opt, mask, ema, range(3, 8))

    model2, opt2, mask2 = _make(gpu, 2)                        # different init, different random masks
    ema2 = ModelEma(model2, decay=0.99)
    with contextlib.redirect_stdout(io.StringIO()):
        assert checkpoint.auto_load_model(args, model2, model2, opt2, None, model_ema=ema2, mask=mask2)
    assert args.start_epoch == 1 and mask2.steps == 3 and mask2.prune_rate == mask_rate_after(3)
    run(model2, opt2, mask2, ema2, range(3, 8))
    assert mask2.prune_rate == mask.prune_rate and mask2.steps == mask.steps == 8
    for n in mask.masks:
        assert torch.equal(mask.masks[n], mask2.masks[n]), n
    for (n, a), b in zip(model.named_parameters(), model2.parameters()):
        assert torch.equal(a, b), n
    for a, b in zip(ema.ema.state_dict().values(), ema2.ema.state_dict().values()):
        assert torch.equal(a, b)
    for a, b in zip(model.parameters(), model2.parameters()):
        for k in ("exp_avg", "exp_avg_sq", "step"):
            assert torch.equal(opt.state[a][k], opt2.state[b][k])


def mask_rate_after(nsteps):
    from slak_amd.sparse_core import CosineDecay
    d = CosineDecay(0.4, 20)
    r = 0.4
    for _ in range(nsteps):
        d.step()
        r = d.get_dr(r)
    return r
