#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE itself.  Run in the build container only
(needs /root/reference; the GPU box never runs this -- it only reads the committed .npz files).

    python tests/golden/make_golden.py

Two families of fixtures:

* ``dwconv_*.npz``  -- what the reference's own tests use as ground truth for the conv op:
  ``F.conv2d(x, w, padding=k//2, groups=C)`` + autograd on torch CPU
  (cutlass/examples/19_large_depthwise_conv2d_torch_extension/test_correctness.py:8-9, :67-90).
  Seeds {0, 42} and ``torch.randn`` inputs as in that file (:20-35); shapes cover the reference's
  square kernels AND the rectangular LoRA kernels of models/SLaK.py:76-80 which the reference never
  pins.  Stored: x, w, dy, y, dx, dw (float32; computed in float64 and rounded once).

* ``mask_*.npz``    -- ``sparse_core.Masking`` + ``funcs.magnitude_prune`` / ``funcs.gradient_growth``
  imported UNMODIFIED from /root/reference and run on CPU (growth='gradient', prune='magnitude',
  redistribution='none', sparse_init='uniform', distributed=False), on a tiny module whose parameter
  names follow models/SLaK.py (so ``--only-L`` name filtering, sparse_core.py:124-127, is exercised).
  Inputs are continuous random values, so no key ties occur at the cut: the
  unstable ``torch.sort`` of the reference is then unambiguous.
"""
import argparse
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# --------------------------------------------------------------------------- conv fixtures
CONV_CASES = [
    # name,            N, C,  H,  W,  kh, kw, seed
    ("sq3_r16",        2, 8, 16, 16,  3,  3, 0),
    ("sq7_r16",        1, 8, 16, 16,  7,  7, 42),
    ("sq13_r16",       2, 4, 16, 16, 13, 13, 0),
    ("sq31_r32",       1, 3, 32, 32, 31, 31, 42),
    ("lora1_51x5_s1",  1, 4, 56, 56, 51,  5, 0),      # cfg-1 stage-1 shape (C reduced)
    ("lora2_5x51_s1",  1, 4, 56, 56,  5, 51, 42),
    ("small_5x5_s1",   2, 4, 56, 56,  5,  5, 0),
    ("lora1_49x5_s2",  2, 6, 28, 28, 49,  5, 42),     # kernel larger than the map
    ("lora2_5x47_s3",  3, 5, 14, 14,  5, 47, 0),
    ("lora1_13x5_s4",  4, 7,  7,  7, 13,  5, 42),
    ("lora2_5x13_s4",  4, 7,  7,  7,  5, 13, 0),
    ("ragged_9x11",    3, 5,  9, 11,  7,  3, 42),     # H != W, odd sizes
    ("one_pixel",      2, 3,  1,  1,  5,  5, 0),
]


def make_conv():
    for name, N, C, H, W, kh, kw, seed in CONV_CASES:
        torch.random.manual_seed(seed)
        x = torch.randn(N, C, H, W)
        conv = nn.Conv2d(C, C, (kh, kw), groups=C, bias=False)      # default init, as the reference tests
        w = conv.weight.detach().clone()
        dy = torch.randn(N, C, H, W)
        xd = x.double().requires_grad_(True)
        wd = w.double().requires_grad_(True)
        y = F.conv2d(xd, wd, None, 1, (kh // 2, kw // 2), 1, C)
        y.backward(dy.double())
        np.savez_compressed(
            os.path.join(HERE, f"dwconv_{name}.npz"),
            x=x.numpy(), w=w.numpy(), dy=dy.numpy(),
            y=y.detach().float().numpy(), dx=xd.grad.float().numpy(), dw=wd.grad.float().numpy(),
            y64=y.detach().numpy(), dx64=xd.grad.numpy(), dw64=wd.grad.numpy(),
            meta=np.array([N, C, H, W, kh, kw, seed]))
        print("wrote dwconv_%s" % name)


# --------------------------------------------------------------------------- mask fixtures
class _ConvBN(nn.Module):
    def __init__(self, C, k):
        super().__init__()
        self.conv = nn.Conv2d(C, C, k, groups=C, bias=False)
        self.bn = nn.BatchNorm2d(C)


class _LK(nn.Module):
    def __init__(self, C, K):
        super().__init__()
        self.LoRA1 = _ConvBN(C, (K, 5))
        self.LoRA2 = _ConvBN(C, (5, K))
        self.small_conv = _ConvBN(C, 5)


class _Block(nn.Module):
    def __init__(self, C, K):
        super().__init__()
        self.large_kernel = _LK(C, K)
        self.norm = nn.LayerNorm(C)
        self.pwconv1 = nn.Linear(C, 4 * C)
        self.pwconv2 = nn.Linear(4 * C, C)
        self.gamma = nn.Parameter(torch.ones(C))


class TinyNet(nn.Module):
    """Parameter names mirror models/SLaK.py:193-201 (``stages.{i}.{j}.large_kernel.LoRA1.conv.weight`` ...)."""

    def __init__(self, C=(6, 10), K=(13, 7)):
        super().__init__()
        self.stages = nn.ModuleList([nn.Sequential(_Block(c, k), _Block(c, k)) for c, k in zip(C, K)])
        self.head = nn.Linear(C[-1], 11)


def _ref_masking(model, optimizer, only_L, sparsity, update_frequency, prune_rate, T_max, seed, growth="gradient"):
    sys.path.insert(0, REF)
    import sparse_core  # noqa: E402  (the reference, unmodified)
    args = types.SimpleNamespace(device="cpu", fix=False, update_frequency=update_frequency, only_L=only_L,
                                 sparse_init="uniform", sparsity=sparsity, distributed=False)
    decay = sparse_core.CosineDecay(prune_rate, T_max)
    torch.manual_seed(seed)
    m = sparse_core.Masking(optimizer, train_loader=None, prune_rate_decay=decay, prune_rate=prune_rate,
                            prune_mode="magnitude", growth_mode=growth, redistribution_mode="none", args=args)
    m.add_module(model)
    return m


def _snap(d):
    return {k: v.detach().cpu().numpy().copy() for k, v in d.items()}


def make_masks():
    import contextlib, io
    # "random" = funcs.random_growth (funcs.py:170-175), the default of main.py:211.  It calls ``.cuda()`` on the numbers it draws from
    # the host generator; this container has no GPU, so for THIS recording Tensor.cuda is the identity (the reference source itself
    # is imported unmodified): the same numbers are compared on the CPU instead of the GPU.
    for tag, only_L, opt_kind in (("all_sgd", False, "sgd"), ("onlyL_adamw", True, "adamw"), ("all_adamw_random", False, "adamw")):
        growth = "random" if tag.endswith("_random") else "gradient"
        _cuda = torch.Tensor.cuda
        if growth == "random":
            torch.Tensor.cuda = lambda self, *a, **k: self
        torch.manual_seed(123)
        model = TinyNet()
        for p in model.parameters():                     # continuous values -> no ties
            p.data = torch.randn_like(p) * 0.05
        if opt_kind == "sgd":
            opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
        else:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.05)
        with contextlib.redirect_stdout(io.StringIO()):
            mask = _ref_masking(model, opt, only_L, sparsity=0.4, update_frequency=3, prune_rate=0.3, T_max=20, seed=7, growth=growth)
        names = list(mask.masks.keys())
        out = {"names": np.array(names), "param_names": np.array([n for n, _ in model.named_parameters()])}
        for n, p in model.named_parameters():
            out[f"w_init/{n}"] = p.detach().numpy().copy()          # AFTER init's apply_mask
        for n in names:
            out[f"m_init/{n}"] = mask.masks[n].numpy().copy()
        nsteps = 7
        g = torch.Generator().manual_seed(99)
        rates = []
        # wrap (NOT modify) the reference's truncate_weights to snapshot the state it sees
        orig_truncate = mask.truncate_weights
        cur = {"step": 0}

        def _wrapped_truncate():
            for n, p in model.named_parameters():
                out[f"wpre{cur['step']}/{n}"] = p.detach().numpy().copy()
            out[f"rate_at{cur['step']}"] = np.array(mask.prune_rate, np.float64)
            orig_truncate()
        mask.truncate_weights = _wrapped_truncate
        for step in range(1, nsteps + 1):
            for n, p in model.named_parameters():
                p.grad = torch.randn(p.shape, generator=g) * 0.1
                out[f"g{step}/{n}"] = p.grad.numpy().copy()
            cur["step"] = step
            with contextlib.redirect_stdout(io.StringIO()):
                mask.step()                                          # sparse_core.py:300-313
            rates.append(mask.prune_rate)
            for n, p in model.named_parameters():
                out[f"w{step}/{n}"] = p.detach().numpy().copy()
            for n in names:
                out[f"m{step}/{n}"] = mask.masks[n].numpy().copy()
            if step % 3 == 0:
                for n in names:
                    out[f"stats{step}/{n}"] = np.array([mask.name2nonzeros[n], mask.name2zeros[n], mask.name2removed[n]], np.float64)
            opt.zero_grad(set_to_none=False)
        out["prune_rates"] = np.array(rates, np.float64)
        out["meta"] = np.array([nsteps, 3, 20], np.int64)            # nsteps, update_frequency, T_max
        out["hyper"] = np.array([0.4, 0.3], np.float64)              # sparsity, prune_rate
        torch.Tensor.cuda = _cuda
        np.savez_compressed(os.path.join(HERE, f"mask_{tag}.npz"), **out)
        print("wrote mask_%s (%d masked tensors)" % (tag, len(names)))

    # single-call fixtures of the two pure functions, including a tie-at-the-cut case that documents the
    # reference's arbitrary behaviour (stored but only compared as a count, not as an index set)
    sys.path.insert(0, REF)
    import funcs  # noqa: E402
    torch.manual_seed(5)
    w = torch.randn(9, 1, 13, 5) * 0.05
    m = (torch.rand(w.shape) < 0.6).float()
    w = w * m
    grad = torch.randn(w.shape)
    fake = types.SimpleNamespace(prune_rate=0.25, name2nonzeros={"t": m.sum().item()},
                                 name2zeros={"t": m.numel() - m.sum().item()})
    pruned = funcs.magnitude_prune(fake, m.clone(), nn.Parameter(w.clone()), "t").float()
    removed = m.sum().item() - pruned.sum().item()
    p = nn.Parameter(w.clone()); p.grad = grad.clone()
    fake.get_gradient_for_weights = lambda weight: weight.grad.clone()
    grown = funcs.gradient_growth(fake, "t", pruned.clone().byte(), math.floor(removed), p).float()
    np.savez_compressed(os.path.join(HERE, "mask_funcs.npz"), w=w.numpy(), m=m.numpy(), grad=grad.numpy(),
                        pruned=pruned.numpy(), grown=grown.numpy(), rate=np.array(0.25), removed=np.array(removed))
    print("wrote mask_funcs")

# --------------------------------------------------------------------------- EMA fixtures
def make_ema():
    """The reference's ModelEma (model_sema.py, imported unmodified) over a run of the reference's Masking + torch AdamW on CPU:
    7 steps, prune-and-grow at steps 3 and 6 (so regrown weights exercise the new_weighs_diff branch), BatchNorm counters advancing
    (int64 entries).  Stored: the model's state dict before every update, the masks, and the EMA state dict after every update;
    plus a dense run (mask=None)."""
    import contextlib, io
    sys.path.insert(0, REF)
    import model_sema  # noqa: E402  (the reference, unmodified)
    for tag, sparse in (("masked", True), ("dense", False)):
        torch.manual_seed(321)
        model = TinyNet()
        for p in model.parameters():
            p.data = torch.randn_like(p) * 0.05
        opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.05)
        mask = None
        if sparse:
            with contextlib.redirect_stdout(io.StringIO()):
                mask = _ref_masking(model, opt, False, sparsity=0.5, update_frequency=3, prune_rate=0.4, T_max=20, seed=11)
        ema = model_sema.ModelEma(model, decay=0.99)
        keys = list(ema.ema.state_dict().keys())
        out = {"keys": np.array(keys), "decay": np.array(0.99, np.float64),
               "mask_names": np.array(list(mask.masks.keys()) if sparse else [], dtype=str)}
        for k, v in ema.ema.state_dict().items():
            out[f"e0/{k}"] = v.numpy().copy()
        g = torch.Generator().manual_seed(17)
        nsteps = 7
        for step in range(1, nsteps + 1):
            for n, p in model.named_parameters():
                p.grad = torch.randn(p.shape, generator=g) * 0.1
            with contextlib.redirect_stdout(io.StringIO()):
                mask.step() if sparse else opt.step()
            for m in model.modules():                                   # what a training forward does to BatchNorm buffers
                if isinstance(m, nn.BatchNorm2d):
                    m.num_batches_tracked += 1
                    m.running_mean += torch.randn(m.running_mean.shape, generator=g) * 0.01
                    m.running_var *= 1.0 + 0.01 * torch.rand(m.running_var.shape, generator=g)
            for k, v in model.state_dict().items():
                out[f"w{step}/{k}"] = v.numpy().copy()
            if sparse:
                for n in mask.masks:
                    out[f"m{step}/{n}"] = mask.masks[n].numpy().copy()
            ema.update(model, mask)                                     # model_sema.py:67-91
            for k, v in ema.ema.state_dict().items():
                out[f"e{step}/{k}"] = v.numpy().copy()
        out["meta"] = np.array([nsteps], np.int64)
        np.savez_compressed(os.path.join(HERE, f"ema_{tag}.npz"), **out)
        print("wrote ema_%s (%d entries)" % (tag, len(keys)))

# --------------------------------------------------------------------------- SNIP fixture (sparse_core.py:11-47; --sparse_init snip)
def _snip_net():
    torch.manual_seed(77)
    return nn.Sequential(nn.Conv2d(3, 6, 3, padding=1), nn.ReLU(), nn.Flatten(), nn.Linear(6 * 8 * 8, 12), nn.ReLU(), nn.Linear(12, 5))


def make_snip():
    sys.path.insert(0, REF)
    import sparse_core  # noqa: E402  (the reference, unmodified)
    net = _snip_net()
    g = torch.Generator().manual_seed(3)
    images, labels = torch.randn(16, 3, 8, 8, generator=g), torch.randint(0, 5, (16,), generator=g)
    masks = {n: torch.zeros_like(p) for n, p in net.named_parameters() if p.dim() in (2, 4)}
    args = types.SimpleNamespace(distributed=False)
    sp = sparse_core.SNIP(net, 0.6, [(images, labels)], torch.device("cpu"), masks, args)
    np.savez_compressed(os.path.join(HERE, "snip_small.npz"), images=images.numpy(), labels=labels.numpy(),
                        names=np.array(list(masks.keys())), sparsities=np.array(sp, np.float64), keep_ratio=np.array(0.6))
    print("wrote snip_small", sp)

# --------------------------------------------------------------------------- ERK init fixture (sparse_core.py:184-262; --sparse_init ERK)
def make_erk():
    import contextlib, io
    sys.path.insert(0, REF)
    import sparse_core  # noqa: E402
    out = {}
    for tag, sparsity in (("s40", 0.4), ("s90", 0.9)):               # 0.9: some layers have to stay dense (the epsilon loop iterates)
        torch.manual_seed(123)
        model = TinyNet()
        args = types.SimpleNamespace(device="cpu", fix=False, update_frequency=3, only_L=False, sparse_init="ERK", sparsity=sparsity,
                                     distributed=False)
        torch.manual_seed(9)
        with contextlib.redirect_stdout(io.StringIO()):
            m = sparse_core.Masking(torch.optim.SGD(model.parameters(), lr=0.1), train_loader=None,
                                    prune_rate_decay=sparse_core.CosineDecay(0.3, 20), prune_rate=0.3, prune_mode="magnitude",
                                    growth_mode="gradient", redistribution_mode="none", args=args)
            m.add_module(model)
        out[f"{tag}/names"] = np.array(list(m.masks.keys()))
        for n, v in m.masks.items():
            out[f"{tag}/m/{n}"] = v.numpy().copy()
        out[f"{tag}/sparsity"] = np.array(sparsity)
    np.savez_compressed(os.path.join(HERE, "erk_init.npz"), **out)
    print("wrote erk_init")


# --------------------------------------------------------------------------- full-model fixture (SURVEY 8c "full-model oracle")
MODEL_CFG = dict(in_chans=3, num_classes=10, depths=[1, 1, 2, 1], dims=[8, 16, 32, 48], drop_path_rate=0.0,
                 layer_scale_init_value=0.5, kernel_size=[31, 29, 13, 7, 5], Decom=True, bn=True)
MODEL_RES = 128            # stage maps 32, 16, 8, 4: the 28x28-class DMA kernels, the 14x14-class kernels, the small-plane kernels


def _import_reference_slak():
    """models/SLaK.py imported UNMODIFIED.  What it needs and this image lacks is shimmed at the import boundary only:
    ``timm.models.layers.trunc_normal_`` (== torch.nn.init.trunc_normal_, absolute bounds [-2, 2]: timm1/layers/weight_init.py:43-67),
    ``DropPath`` (never instantiated: drop_path_rate = 0 -> nn.Identity, models/SLaK.py:150), ``register_model`` (identity) and
    ``DepthWiseConv2dImplicitGEMM`` = the nn.Conv2d the reference's own smoke test compares it with
    (depthwise_conv2d_implicit_gemm.py:69-82: same weight, padding = k // 2, groups = C)."""
    timm = types.ModuleType("timm"); tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers"); tr = types.ModuleType("timm.models.registry")
    tl.trunc_normal_ = lambda t, mean=0., std=1., a=-2., b=2.: nn.init.trunc_normal_(t, mean, std, a, b)

    class _NoDropPath(nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("fixture uses drop_path_rate = 0")
    tl.DropPath = _NoDropPath
    tr.register_model = lambda f: f
    dw = types.ModuleType("depthwise_conv2d_implicit_gemm")

    class DepthWiseConv2dImplicitGEMM(nn.Conv2d):
        def __init__(self, channels, kernel, bias=False):
            super().__init__(channels, channels, kernel, groups=channels, bias=bias)

        def forward(self, x):
            kh, kw = self.kernel_size
            return F.conv2d(x, self.weight, self.bias, 1, (kh // 2, kw // 2), 1, self.groups)
    dw.DepthWiseConv2dImplicitGEMM = DepthWiseConv2dImplicitGEMM
    saved = {k: sys.modules.get(k) for k in ("timm", "timm.models", "timm.models.layers", "timm.models.registry", "depthwise_conv2d_implicit_gemm")}
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl, "timm.models.registry": tr,
                        "depthwise_conv2d_implicit_gemm": dw})
    import importlib.util
    spec = importlib.util.spec_from_file_location("reference_models_SLaK", os.path.join(REF, "models", "SLaK.py"))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod


def make_model():
    """A narrow SLaK built by the REFERENCE's models/SLaK.py (fp64, CPU): state dict, one training-mode forward + backward (batch
    statistics), the BatchNorm running statistics after it, and an eval-mode forward.  tests/test_model_reference_gpu.py loads the
    state dict into slak_amd.slak_model.SLaK and must reproduce all of it through the HIP path."""
    ref = _import_reference_slak()
    ref.use_sync_bn = False                                   # single process (models/SLaK.py:19)
    torch.manual_seed(42)
    model = ref.SLaK(**MODEL_CFG).double()
    with torch.no_grad():                                     # give every affine parameter and running statistic a non-trivial value
        for n, p in model.named_parameters():
            if n.endswith("bn.weight") or n.endswith("norm.weight"):
                p.add_(0.2 * torch.randn_like(p))
            elif n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
            elif "large_kernel" in n and n.endswith("conv.weight"):
                p.mul_(4.0)                                   # branch outputs of O(1) so the BatchNorms see real variances
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = torch.randn(4, 3, MODEL_RES, MODEL_RES, dtype=torch.float64)
    dlogits = torch.randn(4, MODEL_CFG["num_classes"], dtype=torch.float64)
    model.train()
    logits = model(x)
    (logits * dlogits).sum().backward()
    out = {"x": x.float().numpy(), "dlogits": dlogits.float().numpy(), "logits_train": logits.detach().numpy()}
    for k, v in state0.items():
        out["state0/" + k] = v.numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
    for n, p in model.named_parameters():
        out["grad/" + n] = p.grad.numpy()
    for k, v in model.state_dict().items():
        if "running_" in k:
            out["state1/" + k] = v.numpy()
    model.eval()
    with torch.no_grad():
        out["logits_eval"] = model(x).numpy()
    out["names"] = np.array([n for n, _ in model.named_parameters()])
    out["cfg"] = np.array(repr(dict(MODEL_CFG, res=MODEL_RES)))
    np.savez_compressed(os.path.join(HERE, "model_reference.npz"), **out)
    print("wrote model_reference", sum(v.size for v in out.values() if hasattr(v, "size")), "values")


# --------------------------------------------------------------------------- the reference's step loop
ENGINE_ITERS = 6
ENGINE_BATCH = 4
ENGINE_MARGIN = {1: 9e-4, 2: 2e-3}        # smallest relative gap at a prune / regrow cut the chosen data seed must leave (60 / 20 cuts)
ENGINE_FIRST_SEED = {1: 818, 2: 325}      # where the search starts; found by a search from 100 (9 minutes), margins 9.6e-4 / 2.2e-3
ENGINE_HYPER = dict(opt="adamw", lr=4e-3, min_lr=1e-5, weight_decay=0.05, weight_decay_end=0.01, opt_eps=1e-8, opt_betas=None, momentum=0.9,
                    warmup_epochs=0, warmup_steps=-1, epochs=1, start_epoch=0, clip_grad=None, model_ema_decay=0.9,
                    sparse=True, prune="magnitude", growth="gradient", redistribution="none", prune_rate=0.3, fix=False, update_frequency=2,
                    only_L=True, sparse_init="uniform", sparsity=0.4, distributed=False)


def engine_batches(seed, dtype=torch.float32):
    """The synthetic 'data loader' of the engine fixture: ENGINE_ITERS (samples, targets) pairs drawn from a seeded host generator
    (the same torch build on the GPU box draws the same numbers; the fixture stores a checksum)."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(ENGINE_ITERS):
        x = torch.randn(ENGINE_BATCH, 3, MODEL_RES, MODEL_RES, generator=g)
        t = torch.randint(0, MODEL_CFG["num_classes"], (ENGINE_BATCH,), generator=g)
        out.append((x.to(dtype), t))
    return out


class RecordingCriterion(nn.Module):
    """criterion is an ARGUMENT of train_one_epoch (engine.py:17): this one is nn.CrossEntropyLoss that keeps every value it returned."""

    def __init__(self):
        super().__init__()
        self.ce = nn.CrossEntropyLoss()
        self.values = []

    def forward(self, output, target):
        loss = self.ce(output, target)
        self.values.append(loss.detach().double().cpu().item())
        return loss


def _import_reference_training_stack():
    """engine.py, utils.py, optim_factory.py, sparse_core.py, model_sema.py imported UNMODIFIED; their import boundary (timm, tensorboardX,
    torch._six) is supplied by oracle/ref_modules.py's shims plus name-only stubs for the timm optimizers optim_factory.py:12-21 imports."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import ref_modules
    engine, utils = ref_modules.load_engine()
    stubs = {"timm": types.ModuleType("timm"), "timm.optim": types.ModuleType("timm.optim")}
    for mod, cls in (("adafactor", "Adafactor"), ("adahessian", "Adahessian"), ("adamp", "AdamP"), ("lookahead", "Lookahead"), ("nadam", "Nadam"),
                     ("nvnovograd", "NvNovoGrad"), ("radam", "RAdam"), ("rmsprop_tf", "RMSpropTF"), ("sgdp", "SGDP")):
        m = types.ModuleType("timm.optim." + mod)
        setattr(m, cls, type(cls, (), {}))
        stubs["timm.optim." + mod] = m
    saved = {k: sys.modules.get(k) for k in stubs}
    sys.modules.update(stubs)
    sys.path.insert(0, REF)
    try:
        import importlib.util
        spec = importlib.util.spec_from_file_location("reference_optim_factory", os.path.join(REF, "optim_factory.py"))
        optim_factory = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(optim_factory)
        import sparse_core, model_sema, funcs  # noqa: E402
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return engine, utils, optim_factory, sparse_core, model_sema, funcs


def _engine_run(update_freq, data_seed, stack):
    import contextlib, io
    engine, utils, optim_factory, sparse_core, model_sema, funcs = stack
    ref = _import_reference_slak()
    ref.use_sync_bn = False
    args = types.SimpleNamespace(device="cpu", **ENGINE_HYPER)
    torch.manual_seed(42)
    model = ref.SLaK(**MODEL_CFG).double()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bn.weight") or n.endswith("norm.weight"):
                p.add_(0.2 * torch.randn_like(p))
            elif n.endswith(".bias"):
                p.add_(0.05 * torch.randn_like(p))
            elif "large_kernel" in n and n.endswith("conv.weight"):
                p.mul_(4.0)
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sink = io.StringIO()
    margins = []

    def _watch_prune(masking, mask, weight, name):                        # funcs.py:107-114, called through the reference's own table
        num_remove = math.ceil(masking.prune_rate * masking.name2nonzeros[name])
        k = math.ceil(masking.name2zeros[name] + num_remove)
        x, _ = torch.sort(torch.abs(weight.data.view(-1)))
        if 0 < k < x.numel() and num_remove > 0:
            margins.append(float((x[k] - x[k - 1]) / x[k]))
        return orig_prune(masking, mask, weight, name)

    def _watch_growth(masking, name, new_mask, total_regrowth, weight):   # funcs.py:196-205
        grad = masking.get_gradient_for_weights(weight) * (new_mask == 0).float()
        y, _ = torch.sort(torch.abs(grad).flatten(), descending=True)
        t = int(total_regrowth)
        if 0 < t < y.numel():
            margins.append(float((y[t - 1] - y[t]) / y[t - 1]))
        return orig_growth(masking, name, new_mask, total_regrowth, weight)
    orig_prune, orig_growth = sparse_core.prune_funcs["magnitude"], sparse_core.growth_funcs["gradient"]
    sparse_core.prune_funcs["magnitude"], sparse_core.growth_funcs["gradient"] = _watch_prune, _watch_growth
    sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None                          # engine.py:90 on a host without a GPU: nothing to wait for
    try:
        with contextlib.redirect_stdout(sink):
            # ---- main.py:339-347, 384-425, in main.py's order
            model_ema = model_sema.ModelEma(model, decay=args.model_ema_decay, device="", resume="")
            steps_per_epoch = ENGINE_ITERS // update_freq
            optimizer = optim_factory.create_optimizer(args, model, skip_list=None, get_num_layer=None, get_layer_scale=None)
            lr_values = utils.cosine_scheduler(args.lr, args.min_lr, args.epochs, steps_per_epoch, warmup_epochs=args.warmup_epochs, warmup_steps=args.warmup_steps)
            wd_values = utils.cosine_scheduler(args.weight_decay, args.weight_decay_end, args.epochs, steps_per_epoch)
            criterion = RecordingCriterion()
            loader = engine_batches(data_seed, torch.float64)
            torch.manual_seed(7)                                           # the uniform mask init draws from the host generator (sparse_core.py:180)
            decay = sparse_core.CosineDecay(args.prune_rate, int(steps_per_epoch * args.epochs), init_step=int(steps_per_epoch) * args.start_epoch)
            mask = sparse_core.Masking(optimizer, train_loader=loader, prune_mode=args.prune, prune_rate_decay=decay, growth_mode=args.growth,
                                       redistribution_mode=args.redistribution, args=args)
            mask.add_module(model)
            m_init = {n: m.numpy().copy() for n, m in mask.masks.items()}
            stats = engine.train_one_epoch(model, criterion, loader, optimizer, torch.device("cpu"), 0, None, args.clip_grad, model_ema, None,
                                           log_writer=None, wandb_logger=None, start_steps=0, lr_schedule_values=lr_values, wd_schedule_values=wd_values,
                                           num_training_steps_per_epoch=steps_per_epoch, update_freq=update_freq, use_amp=False, mask=mask)
    finally:
        torch.cuda.synchronize = sync
        sparse_core.prune_funcs["magnitude"], sparse_core.growth_funcs["gradient"] = orig_prune, orig_growth
    out = {"losses": np.array(criterion.values, np.float64), "mask_steps": np.array(mask.steps), "prune_rate": np.array(mask.prune_rate, np.float64),
           "mask_names": np.array(list(mask.masks.keys())), "stat_loss": np.array(stats["loss"], np.float64), "stat_lr": np.array(stats["lr"], np.float64),
           "stat_weight_decay": np.array(stats["weight_decay"], np.float64), "update_freq": np.array(update_freq), "data_seed": np.array(data_seed),
           "data_checksum": np.array(sum(float(x.double().sum()) for x, _ in loader), np.float64), "min_margin": np.array(min(margins), np.float64),
           "n_cuts": np.array(len(margins)), "hyper": np.array(repr(ENGINE_HYPER)), "cfg": np.array(repr(dict(MODEL_CFG, res=MODEL_RES)))}
    for k, v in state0.items():
        out["state0/" + k] = v.numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
    for n, m in m_init.items():
        out["m_init/" + n] = np.packbits(m.astype(np.uint8).reshape(-1))
    for n, m in mask.masks.items():
        out["m_final/" + n] = np.packbits(m.numpy().astype(np.uint8).reshape(-1))
    for k, v in model.state_dict().items():
        out["w_final/" + k] = v.numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
    for k, v in model_ema.ema.state_dict().items():
        out["ema_final/" + k] = v.numpy().astype(np.float32) if v.dtype.is_floating_point else v.numpy()
    return out


def make_engine():
    """The reference's OWN step loop: engine.train_one_epoch (engine.py:17-140) driving the reference's models/SLaK.py (narrow, fp64), the optimizer of
    optim_factory.create_optimizer, sparse_core.Masking built as main.py:421-425 builds it, model_sema.ModelEma -- all imported unmodified, CPU.
    tests/test_reference_engine_gpu.py runs the SAME unmodified train_one_epoch on the product (mirror model on libslak_hip.so, slak_amd Masking / MaskedAdamW /
    ModelEma) and must reproduce the losses, the masks bit for bit, the weights and the EMA.  The data seed is the first whose prune / regrow cuts all have a
    relative gap >= ENGINE_MARGIN (9e-4 / 2e-3) between the last element taken and the first one left (fp32-vs-fp64 rounding of a gradient cannot move an element across)."""
    stack = _import_reference_training_stack()
    for uf in (1, 2):
        best = None
        for seed in range(ENGINE_FIRST_SEED[uf], 1100):
            out = _engine_run(uf, seed, stack)
            if best is None or out["min_margin"] > best["min_margin"]:
                best = out
                print("  update_freq %d seed %d: %d cuts, min margin %.2e, losses %s" % (uf, seed, out["n_cuts"], out["min_margin"], np.round(out["losses"], 4)))
            if out["min_margin"] >= ENGINE_MARGIN[uf]:
                break
        np.savez_compressed(os.path.join(HERE, f"engine_uf{uf}.npz"), **best)
        print("wrote engine_uf%d (seed %d, min margin %.2e, mask.steps %d)" % (uf, best["data_seed"], best["min_margin"], best["mask_steps"]))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", choices=["conv", "mask", "ema", "snip", "erk", "model", "engine"], default=None)
    a = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    if a.only in (None, "conv"):
        make_conv()
    if a.only in (None, "mask"):
        make_masks()
    if a.only in (None, "ema"):
        make_ema()
    if a.only in (None, "snip"):
        make_snip()
    if a.only in (None, "erk"):
        make_erk()
    if a.only in (None, "model"):
        make_model()
    if a.only in (None, "engine"):
        make_engine()
