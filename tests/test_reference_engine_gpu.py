"""the reference's OWN step loop drives the product.

``engine.train_one_epoch`` (/root/reference/engine.py:17-140) is loaded UNMODIFIED (oracle/ref_modules.load_engine: source in the build container, the
bytecode under oracle/_ref/ on the GPU box; only its import boundary -- timm, tensorboardX, torch._six -- is supplied) and called with

    model      slak_amd.slak_model.SLaK            (the caller-side mirror of models/SLaK.py on libslak_hip.so)
    optimizer  slak_amd.optim_factory.create_optimizer(args, model)     -> MaskedAdamW           (main.py:378-381)
    mask       slak_amd.sparse_core.Masking built with main.py:421-425's expression, word for word
    model_ema  slak_amd.model_sema.ModelEma(model, decay=..., device='', resume='')                (main.py:339-346)

exactly as main.py:443-450 calls it (positional order and keyword names included).  The checker: tests/golden/engine_uf{1,2}.npz, the same call made in the
build container on the reference's own model / optimizer factory / Masking / ModelEma in fp64 (tests/golden/make_golden.py --only engine).
Asserted: ``mask.step()`` fired on every optimizer step (engine.py:82-83), the prune-and-grow rounds happened, the masks are BIT-EXACT, per-iteration losses,
the returned statistics, the final weights and the EMA agree to fp32-vs-fp64 accuracy, and losses / masks / weights equal those of a hand-written loop of the
same step (bench.py's) on the same product."""
import ast
import contextlib
import io
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
sys.path.insert(0, GOLDEN)


def _engine():
    from oracle import ref_modules
    if not (ref_modules.available("reference_engine") and ref_modules.available("reference_utils")):
        pytest.fail("oracle/_ref/reference_engine.pyc is missing: run __graft_entry__.build() in the build container")
    return ref_modules.load_engine()


def _args(g, gpu):
    hyper = ast.literal_eval(str(g["hyper"]))
    return types.SimpleNamespace(device=str(gpu), **hyper)


def _model(g, gpu, fused=False, lowp=False):
    import slak_amd.slak_model as M
    from slak_amd import block_ops
    cfg = ast.literal_eval(str(g["cfg"]))
    cfg.pop("res")
    M.use_sync_bn = False
    M.Block.fused_block = False
    M.Block.fused_tail = fused
    M.ReparamLargeKernelConv.fused_bn = fused
    M.ReparamLargeKernelConv.fused_tri = fused
    M.LayerNorm.fused_cf = fused
    block_ops.cache_lowp_weights = False
    m = M.SLaK(lowp_dwconv=lowp, **cfg)
    keys = [k[len("state0/"):] for k in g if k.startswith("state0/")]
    m.load_state_dict({k: torch.from_numpy(g["state0/" + k]) for k in keys}, strict=True)
    return m.to(gpu)


def _construct(g, gpu, utils, uf, model):
    """main.py:339-347, 378-425 on the product's classes; the expressions are main.py's."""
    from make_golden import ENGINE_ITERS, RecordingCriterion, engine_batches
    from slak_amd.model_sema import ModelEma
    from slak_amd.optim_factory import MaskedAdamW, create_optimizer
    from slak_amd.sparse_core import CosineDecay, Masking
    args = _args(g, gpu)
    model_ema = ModelEma(model, decay=args.model_ema_decay, device='', resume='')
    num_training_steps_per_epoch = ENGINE_ITERS // uf
    optimizer = create_optimizer(args, model, skip_list=None, get_num_layer=None, get_layer_scale=None)
    assert isinstance(optimizer, MaskedAdamW)
    lr_schedule_values = utils.cosine_scheduler(args.lr, args.min_lr, args.epochs, num_training_steps_per_epoch,
                                                warmup_epochs=args.warmup_epochs, warmup_steps=args.warmup_steps)
    wd_schedule_values = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs, num_training_steps_per_epoch)
    criterion = RecordingCriterion()
    data_loader_train = engine_batches(int(g["data_seed"]))
    assert abs(sum(float(x.double().sum()) for x, _ in data_loader_train) - float(g["data_checksum"])) < 1e-2, "host generator drew other numbers"
    torch.manual_seed(7)
    decay = CosineDecay(args.prune_rate, int(num_training_steps_per_epoch*args.epochs), init_step= int(num_training_steps_per_epoch)*(args.start_epoch))
    mask = Masking(optimizer, train_loader=data_loader_train, prune_mode=args.prune, prune_rate_decay=decay, growth_mode=args.growth, redistribution_mode=args.redistribution, args=args)
    mask.add_module(model)
    return args, model_ema, optimizer, lr_schedule_values, wd_schedule_values, criterion, data_loader_train, mask, num_training_steps_per_epoch


def _unpack(packed, shape):
    return np.unpackbits(packed)[:int(np.prod(shape))].reshape(shape).astype(np.float32)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


@pytest.mark.parametrize("uf", [1, 2])
def test_reference_train_one_epoch_drives_the_product(uf, gpu):
    engine, utils = _engine()
    g = load_golden("engine_uf%d" % uf)
    assert int(g["update_freq"]) == uf
    model = _model(g, gpu)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        args, model_ema, optimizer, lr_values, wd_values, criterion, loader, mask, steps_per_epoch = _construct(g, gpu, utils, uf, model)
    names = [str(n) for n in g["mask_names"]]
    assert list(mask.masks.keys()) == names
    for n in names:
        np.testing.assert_array_equal(mask.masks[n].cpu().numpy(), _unpack(g["m_init/" + n], mask.masks[n].shape), err_msg="init mask " + n)
    step_calls, ema_calls = [], []
    orig_step, orig_update = mask.step, model_ema.update
    mask.step = lambda: (step_calls.append(mask.steps), orig_step())[1]
    model_ema.update = lambda m, msk: (ema_calls.append(msk is mask), orig_update(m, msk))[1]
    with contextlib.redirect_stdout(sink):
        # ---- main.py:443-450, verbatim (loss_scaler is unused with use_amp=False: engine.py:68-76 is the other branch)
        train_stats = engine.train_one_epoch(
            model, criterion, loader, optimizer,
            gpu, 0, None, args.clip_grad, model_ema, None,
            log_writer=None, wandb_logger=None, start_steps=0 * steps_per_epoch,
            lr_schedule_values=lr_values, wd_schedule_values=wd_values,
            num_training_steps_per_epoch=steps_per_epoch, update_freq=uf,
            use_amp=False, mask=mask
        )
    log = sink.getvalue()
    assert "Averaged stats:" in log and "Epoch: [0]" in log                     # utils.MetricLogger ran (engine.py:24-32, :139)
    # engine.py:80-88: one mask.step() and one EMA update per optimizer step, none in between
    assert step_calls == list(range(steps_per_epoch)) and mask.steps == int(g["mask_steps"]) == steps_per_epoch
    assert ema_calls == [True] * steps_per_epoch
    assert mask.prune_rate == float(g["prune_rate"])
    assert all(p.grad is None or not p.grad.any() for p in model.parameters())   # engine.py:86 zero_grad after the step
    losses = np.array(criterion.values)
    assert np.isfinite(losses).all()
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-4, atol=0)
    assert abs(train_stats["loss"] - float(g["stat_loss"])) <= 2e-4 * float(g["stat_loss"])
    assert train_stats["lr"] == float(g["stat_lr"]) and train_stats["weight_decay"] == float(g["stat_weight_decay"])
    assert 0.0 <= train_stats["class_acc"] <= 1.0
    # masks: bit-exact against the reference Masking driven by the reference engine
    for n in names:
        got = mask.masks[n].cpu().numpy()
        want = _unpack(g["m_final/" + n], got.shape)
        assert np.array_equal(got, want), (n, int((got != want).sum()), float(g["min_margin"]))
        assert not np.array_equal(want, _unpack(g["m_init/" + n], got.shape)), "no prune-and-grow round changed " + n
    state = model.state_dict()
    for k, v in state.items():
        if v.dtype.is_floating_point:
            assert _rel(v.double().cpu().numpy(), g["w_final/" + k]) <= 1e-3, k
        else:
            assert np.array_equal(v.cpu().numpy(), g["w_final/" + k]), k
    for k, v in model_ema.ema.state_dict().items():
        if v.dtype.is_floating_point:
            assert _rel(v.double().cpu().numpy(), g["ema_final/" + k]) <= 1e-3, k
        else:
            assert np.array_equal(v.cpu().numpy(), g["ema_final/" + k]), k

    # ---- the same step written by hand (bench.py's loop) on a second instance: identical losses, masks and weights
    model2 = _model(g, gpu)
    with contextlib.redirect_stdout(sink):
        args, ema2, opt2, lr_values, wd_values, crit2, loader2, mask2, _ = _construct(g, gpu, utils, uf, model2)
        model2.train(True)
        opt2.zero_grad()
        for i, (samples, targets) in enumerate(loader2):
            it = i // uf
            for group in opt2.param_groups:
                group["lr"] = lr_values[it] * group["lr_scale"]
                if group["weight_decay"] > 0:
                    group["weight_decay"] = wd_values[it]
            loss = crit2(model2(samples.to(gpu)), targets.to(gpu)) / uf
            loss.backward()
            if (i + 1) % uf == 0:
                mask2.step()
                opt2.zero_grad()
                ema2.update(model2, mask2)
    # (not bit for bit: in this fp32 composition the stem / downsample convolutions and the pointwise GEMMs are torch's MIOpen / hipBLASLt kernels, whose
    # split reductions are not run-to-run reproducible -- two runs of the SAME loop differ in the last digits too)
    np.testing.assert_allclose(crit2.values, criterion.values, rtol=2e-6, atol=0)
    for n in names:
        assert torch.equal(mask.masks[n], mask2.masks[n]), n
    for (k, v), (_, v2) in zip(model.state_dict().items(), model2.state_dict().items()):
        if v.dtype.is_floating_point:
            assert _rel(v.double().cpu().numpy(), v2.double().cpu().numpy()) <= 1e-5, k
        else:
            assert torch.equal(v, v2), k


def test_reference_engine_on_the_fused_bf16_product(gpu):
    """The same unmodified loop over the configuration bench.py measures: every fused block op on, bf16 dw convs and GEMMs under autocast.  engine.py has
    no bf16 switch (use_amp means fp16 + GradScaler, engine.py:50-53), so the autocast region is the model's own forward (a forward pre/post hook pair --
    the engine is untouched).  Losses stay within the bf16 tolerance of the reference's fp64 run for the steps before the first prune-and-grow round, and
    ``mask.step()`` keeps firing; mask sets are NOT compared here (bf16 gradients may order two near-equal candidates differently)."""
    engine, utils = _engine()
    g = load_golden("engine_uf1")
    model = _model(g, gpu, fused=True, lowp=True)
    try:
        ctx = []

        def _enter(module, inputs):
            ctx.append(torch.autocast("cuda", dtype=torch.bfloat16))
            ctx[-1].__enter__()

        def _exit(module, inputs, out):
            ctx.pop().__exit__(None, None, None)
            return out.float()
        model.register_forward_pre_hook(_enter)
        model.register_forward_hook(_exit)
        sink = io.StringIO()
        with contextlib.redirect_stdout(sink):
            args, model_ema, optimizer, lr_values, wd_values, criterion, loader, mask, steps_per_epoch = _construct(g, gpu, utils, 1, model)
            train_stats = engine.train_one_epoch(
                model, criterion, loader, optimizer,
                gpu, 0, None, args.clip_grad, model_ema, None,
                log_writer=None, wandb_logger=None, start_steps=0,
                lr_schedule_values=lr_values, wd_schedule_values=wd_values,
                num_training_steps_per_epoch=steps_per_epoch, update_freq=1,
                use_amp=False, mask=mask
            )
    finally:
        _model(g, gpu)                                                            # class flags back to the reference composition
    assert mask.steps == steps_per_epoch
    losses = np.array(criterion.values)
    assert np.isfinite(losses).all() and np.isfinite(train_stats["loss"])
    np.testing.assert_allclose(losses[:2], g["losses"][:2], rtol=1e-2)            # before the first prune-and-grow round (update_frequency = 2)
    np.testing.assert_allclose(losses, g["losses"], rtol=5e-2)
    for n, m in mask.masks.items():                                               # prune-and-grow keeps every tensor's budget (sparse_core.py:335-357) ...
        init = _unpack(g["m_init/" + n], m.shape)
        assert int(m.sum().item()) == int(init.sum()), n
        assert not np.array_equal(m.cpu().numpy(), init), n                       # ... and moved some of it


def test_reference_engine_amp_branch_on_the_product(gpu):
    """engine.py:50-53, 66-76: the use_amp=True branch -- fp16 autocast (torch.cuda.amp.autocast's default dtype), utils.NativeScalerWithGradNormCount (GradScaler:
    scale, unscale_, grad norm, scaler.step(optimizer)) -- on the product's model and MaskedAdamW.  The reference's op receives fp32 there
    (depthwise_conv2d_implicit_gemm.py:16 cast_inputs=torch.float32) and so does the mirror (lowp_dwconv=False: the exact fp32 kernels); this branch NEVER calls
    mask.step() (the reference's AMP path trains dense), which the test pins: masks unchanged, mask.steps == 0, the optimizer stepped through the scaler."""
    engine, utils = _engine()
    g = load_golden("engine_uf1")
    model = _model(g, gpu)
    sink = io.StringIO()
    with contextlib.redirect_stdout(sink):
        args, model_ema, optimizer, lr_values, wd_values, criterion, loader, mask, steps_per_epoch = _construct(g, gpu, utils, 1, model)
        loss_scaler = utils.NativeScalerWithGradNormCount()                      # main.py:385
        w0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
        m0 = {n: m.clone() for n, m in mask.masks.items()}
        train_stats = engine.train_one_epoch(
            model, criterion, loader, optimizer,
            gpu, 0, loss_scaler, args.clip_grad, model_ema, None,
            log_writer=None, wandb_logger=None, start_steps=0,
            lr_schedule_values=lr_values, wd_schedule_values=wd_values,
            num_training_steps_per_epoch=steps_per_epoch, update_freq=1,
            use_amp=True, mask=mask
        )
    losses = np.array(criterion.values)
    assert np.isfinite(losses).all() and "grad_norm" in train_stats
    np.testing.assert_allclose(losses[0], g["losses"][0], rtol=5e-3)              # same weights, fp16 GEMMs: the first loss is the fp64 run's to fp16 accuracy
    np.testing.assert_allclose(losses, g["losses"], rtol=0.15)                    # (dense from here on: the recorded run prunes and regrows)
    assert mask.steps == 0 and all(torch.equal(mask.masks[n], m0[n]) for n in m0)  # engine.py:68-76 has no mask.step()
    # GradScaler starts at 2^16: iterations whose fp16 gradients overflow are skipped and halve the scale (their grad norm is inf, as timm logs it), the others step
    # MaskedAdamW through scaler.step(optimizer): steps taken + overflows = iterations, and the scale says how many overflowed
    some = next(p_ for p_ in model.parameters() if p_ in optimizer.state)
    taken = int(float(optimizer.state[some]["step"]))
    overflows = len(losses) - taken
    assert 0 <= overflows <= len(losses) and loss_scaler.state_dict()["scale"] == 65536.0 * 0.5 ** overflows
    moved = [k for k, v in model.state_dict().items() if v.dtype.is_floating_point and "running" not in k and not torch.equal(v, w0[k])]
    assert (len(moved) > 50) == (taken > 0), (len(moved), taken)                  # scaler.step(optimizer) ran the one-launch AdamW whenever the gradients were finite
