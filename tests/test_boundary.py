"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol the header
declares, the product never routes through the oracle or a CPU fallback, and the Python mirrors keep the
reference's surface (depthwise_conv2d_implicit_gemm.py:11-66, sparse_core.py:80-132)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from slak_amd import build
    return build.build()


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "slak_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(slak_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built_lib):
    syms = _header_symbols()
    assert len(syms) >= 17
    L = ctypes.CDLL(built_lib)
    for s in syms:
        assert hasattr(L, s), "libslak_hip.so does not export %s" % s
    from slak_amd import _lib
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding and header disagree"
    lib = _lib.lib()
    assert lib.slak_version() == 8                          # SLAK_ABI_VERSION of include/slak_hip.h
    assert lib.slak_status_string(0) == b"ok" and lib.slak_status_string(3) == b"workspace missing or too small"
    # pure host-side argument validation (no GPU needed): status codes instead of exit()
    assert lib.slak_dwconv2d_forward(None, 0, None, 0, None, 0, 1, 1, 1, 1, 3, 3, None, 0, None) == 1
    assert lib.slak_set_conv_algo(99) == 1
    assert lib.slak_dwconv2d_workspace_bytes(0, 128, 96, 56, 56, 51, 5, 2) > 0


def test_fp32_matrix_core_switches_are_host_state(built_lib):
    """slak_set_fp32_matrix_cores (process-wide, off by default) and the per-thread override the op module uses under autocast: pure host
    state, nests through the Python context manager, does not leak into other threads."""
    import threading
    from slak_amd import _lib, ops
    L = _lib.lib()
    eff, glob = L.slak_get_fp32_matrix_cores_effective, L.slak_get_fp32_matrix_cores
    assert glob() == 0 and eff() == 0
    with ops.fp32_matrix_cores(True):
        assert eff() == 1 and glob() == 0                          # the override is this thread's; the process-wide switch is untouched
        with ops.fp32_matrix_cores(False):
            assert eff() == 0
        assert eff() == 1
        seen = []
        t = threading.Thread(target=lambda: seen.append(eff()))
        t.start(); t.join()
        assert seen == [0]                                         # another thread follows the process-wide setting
        # a save / restore pair INSIDE an override must not write the override into the process-wide switch
        prev = ops.allow_fp32_matrix_cores(False)
        assert prev is False
        ops.allow_fp32_matrix_cores(prev)
        assert glob() == 0
    assert eff() == 0 and glob() == 0
    prev = ops.allow_fp32_matrix_cores(True)
    try:
        assert prev is False and glob() == 1 and eff() == 1
        with ops.fp32_matrix_cores(False):
            assert eff() == 0 and glob() == 1
            assert ops.allow_fp32_matrix_cores(True) is True       # reports the switch, not the override
    finally:
        ops.allow_fp32_matrix_cores(prev)
    assert glob() == 0
    assert L.slak_set_fp32_matrix_cores_thread(7, None) == 1       # SLAK_ERR_INVALID_ARG
    # shapes AUTO takes to the matrix cores need scratch for the packed filter fragments, per chunk of five rows
    assert L.slak_dwconv2d_workspace_bytes(0, 8, 16, 32, 32, 31, 31, 2) >= 7 * L.slak_dwconv2d_workspace_bytes(0, 8, 16, 32, 32, 5, 31, 2) // 2


def test_library_contains_gfx950_code(built_lib):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", built_lib], capture_output=True, text=True).stdout
    if not out:
        pytest.skip("llvm-objdump --offloading unavailable")
    assert "gfx950" in out


def test_product_never_imports_the_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "slak_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M) or "oracle/" in txt and f.endswith((".hip", ".h")):
                    bad.append(f)
    assert not bad, "product files reference the oracle: %s" % bad


def test_no_cpu_fallback_fails_loudly():
    from slak_amd import DepthWiseConv2dImplicitGEMM
    m = DepthWiseConv2dImplicitGEMM(4, (13, 5))
    with pytest.raises(RuntimeError):
        m(torch.randn(1, 4, 8, 8))                        # CPU tensor: refuse, do not silently run F.conv2d
    from slak_amd.sparse_core import CosineDecay, Masking
    import types
    net = nn.Sequential(nn.Linear(8, 8))
    args = types.SimpleNamespace(device="cpu", fix=False, update_frequency=2, only_L=False, sparse_init="uniform", sparsity=0.5, distributed=False)
    from slak_amd._lib import SlakHipError
    with pytest.raises(SlakHipError):
        mk = Masking(torch.optim.SGD(net.parameters(), lr=0.1), None, CosineDecay(0.3, 10), prune_rate=0.3,
                     prune_mode="magnitude", growth_mode="gradient", redistribution_mode="none", args=args)
        mk.add_module(net)                                # apply_mask on CPU params must raise, not fall back
    with pytest.raises(NotImplementedError):
        Masking(None, None, None, growth_mode="momentum", args=args).init_growth_prune_and_redist()   # not on this path: refuse by name


def test_module_surface_matches_reference():
    from slak_amd import DepthWiseConv2dImplicitGEMM
    import slak_amd.depthwise_conv2d_implicit_gemm as mod
    assert mod.__all__ == ["DepthWiseConv2dImplicitGEMM"]
    m = DepthWiseConv2dImplicitGEMM(6, (51, 5), bias=True)
    assert isinstance(m, nn.Conv2d) and m.groups == 6 and m.padding == (0, 0)
    assert m.weight.shape == (6, 1, 51, 5) and m.bias.shape == (6,)
    ref = nn.Conv2d(6, 6, (51, 5), padding=(25, 2), groups=6, bias=True)
    ref.load_state_dict(m.state_dict())                   # identical state-dict keys / shapes
    import slak_amd.ops as ops
    for n in ("forward_fp32", "backward_data_fp32", "backward_filter_fp32", "forward_fp16", "backward_data_fp16", "backward_filter_fp16"):
        assert callable(getattr(ops, n))                  # frontend.cpp:3-16


def test_cosine_decay_matches_closed_form():
    import math, warnings
    from slak_amd.sparse_core import CosineDecay
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        d = CosineDecay(0.3, 50, init_step=3)
        for s in range(4, 10):
            d.step()
            want = 0.005 + (0.3 - 0.005) * (1 + math.cos(math.pi * s / 50)) / 2
            assert abs(d.get_dr(0.3) - want) < 1e-12


def test_optimizer_and_ema_have_no_cpu_fallback():
    """SURVEY 8f-3 product path: CPU tensors are refused loudly instead of being routed through torch ops or the oracle."""
    import torch
    from slak_amd import _lib
    from slak_amd.model_sema import ModelEma
    from slak_amd.optim_factory import MaskedAdamW
    m = torch.nn.Linear(4, 4)
    ema = ModelEma(m)
    with pytest.raises(_lib.SlakHipError):
        ema.update(m, None)
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(_lib.SlakHipError):
        MaskedAdamW([p]).step()


def test_parameter_groups_follow_the_reference_rule():
    """optim_factory.py:73-112: 1-D parameters and biases get no weight decay; layer ids of optim_factory.py:32-59."""
    import contextlib, io
    import torch
    from slak_amd import optim_factory as OF
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.LayerNorm(4))
    with contextlib.redirect_stdout(io.StringIO()):
        groups = OF.get_parameter_groups(net, weight_decay=0.05)
    assert [(g["weight_decay"], len(g["params"]), g["lr_scale"]) for g in groups] == [(0.05, 1, 1.0), (0.0, 3, 1.0)]
    ids = [OF.get_num_layer_for_convnext(n) for n in ("downsample_layers.0.0.weight", "downsample_layers.2.1.bias", "stages.0.2.gamma",
                                                      "stages.2.7.pwconv1.weight", "stages.3.0.norm.bias", "head.weight")]
    assert ids == [0, 3, 1, 5, 12, 13]
    assigner = OF.LayerDecayValueAssigner([0.5 ** (13 - i) for i in range(14)])
    with contextlib.redirect_stdout(io.StringIO()):
        groups = OF.get_parameter_groups(net, 0.05, (), lambda n: 13, assigner.get_scale)
    assert all(g["lr_scale"] == 1.0 for g in groups)


def test_mask_bit_packing_round_trip():
    """Masking.state_dict stores masks 1 bit per weight in numpy.packbits order."""
    import numpy as np
    import torch
    from slak_amd.sparse_core import Masking
    g = torch.Generator().manual_seed(0)
    for shape in [(1,), (7,), (8,), (3, 1, 51, 5), (96, 384), (5, 1, 5, 5)]:
        m = (torch.rand(shape, generator=g) < 0.4).float()
        p = Masking.pack_mask(m)
        np.testing.assert_array_equal(p.numpy(), np.packbits(m.numpy().reshape(-1).astype(np.uint8)))
        assert torch.equal(Masking.unpack_mask(p, shape), m)


def test_model_ema_entries_are_the_state_dict(tmp_path):
    """ModelEma walks parameters and persistent buffers itself (no state_dict() call per step): same keys, same order, same tensors,
    for a plain and for a wrapped ('module.' prefixed) model; a key the model lacks raises KeyError like the reference's msd[k]."""
    import torch
    import slak_amd.slak_model as M
    from slak_amd.model_sema import ModelEma
    net = M.SLaK(in_chans=3, num_classes=5, depths=[1, 1, 1, 1], dims=[8, 16, 24, 32], kernel_size=[7, 7, 5, 5, 3], Decom=True, bn=True)
    ent = ModelEma._entries(net)
    sd = net.state_dict()
    assert [k for k, _ in ent] == list(sd.keys())
    assert all(t.data_ptr() == sd[k].data_ptr() for k, t in ent)

    class Wrap(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.module = m
    ema = ModelEma(net, decay=0.5)
    assert not ema.ema_has_module and not any(p.requires_grad for p in ema.ema.parameters())
    with pytest.raises(Exception):                       # CPU tensors: refused (SlakHipError), but only AFTER the key mapping worked
        ema.update(Wrap(net), None)
    assert [k for k, _, _ in ema._tensors[1]][:2] == ["module." + k for k in list(sd.keys())[:2]]
    with pytest.raises(KeyError):
        ModelEma(net).update(torch.nn.Linear(2, 2), None)


def test_checkpoint_round_trip_on_cpu(tmp_path):
    """slak_amd.checkpoint mirrors utils.save_model / auto_load_model1 (utils.py:447-512): file name, keys, pruning of old files,
    auto-resume from the newest checkpoint; plus the 'mask' entry (any object with state_dict / load_state_dict)."""
    import types
    import torch
    from slak_amd import checkpoint

    class FakeMask:
        def __init__(self): self.state = {"steps": 0}
        def state_dict(self): return dict(self.state)
        def load_state_dict(self, s): self.state = dict(s)

    torch.manual_seed(0)
    net, net2 = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
    opt, opt2 = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9), torch.optim.SGD(net2.parameters(), lr=0.1, momentum=0.9)
    net(torch.randn(4, 3)).sum().backward(); opt.step()
    mask, mask2 = FakeMask(), FakeMask()
    mask.state["steps"] = 7
    args = types.SimpleNamespace(output_dir=str(tmp_path), save_ckpt_num=2, save_ckpt_freq=1, resume='', auto_resume=True)
    for epoch in range(4):
        checkpoint.save_model(args, epoch, net, net, opt, None, model_ema=None, mask=mask)
    import os
    assert sorted(os.listdir(tmp_path)) == ["checkpoint-2.pth", "checkpoint-3.pth"]       # older ones pruned (utils.py:464-468)
    ck = torch.load(os.path.join(tmp_path, "checkpoint-3.pth"), weights_only=False)
    assert set(ck.keys()) == {"model", "optimizer", "epoch", "scaler", "args", "mask"}
    assert checkpoint.auto_load_model(args, net2, net2, opt2, None, model_ema=None, mask=mask2)
    assert args.resume.endswith("checkpoint-3.pth") and args.start_epoch == 4 and mask2.state["steps"] == 7
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
    assert torch.equal(opt.state_dict()["state"][0]["momentum_buffer"], opt2.state_dict()["state"][0]["momentum_buffer"])
    empty = types.SimpleNamespace(output_dir=str(tmp_path / "none"), resume='', auto_resume=True)
    assert checkpoint.auto_load_model(empty, net2, net2, opt2, None) is False


def test_create_optimizer_surface():
    """optim_factory.create_optimizer (optim_factory.py:115-199): adamw -> MaskedAdamW over the reference's two parameter groups,
    torch.optim pass-throughs, timm/apex names raise instead of silently substituting."""
    import contextlib, io, types
    import torch
    from slak_amd import optim_factory as OF
    net = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.LayerNorm(4))
    args = types.SimpleNamespace(opt="adamw", lr=4e-3, weight_decay=0.05, momentum=0.9, opt_eps=1e-8, opt_betas=None)
    with contextlib.redirect_stdout(io.StringIO()):
        opt = OF.create_optimizer(args, net)
    assert isinstance(opt, OF.MaskedAdamW) and opt.applies_masks
    assert [(g["weight_decay"], g["lr"], g["lr_scale"]) for g in opt.param_groups] == [(0.05, 4e-3, 1.0), (0.0, 4e-3, 1.0)]
    args.opt = "sgd"
    with contextlib.redirect_stdout(io.StringIO()):
        assert isinstance(OF.create_optimizer(args, net), torch.optim.SGD)
    for name in ("lookahead_adam", "fusedlamb", "adamp"):
        args.opt = name
        with contextlib.redirect_stdout(io.StringIO()), pytest.raises(NotImplementedError):
            OF.create_optimizer(args, net)


def test_masking_accessors():
    """Masking.get_momentum_for_weight (sparse_core.py:362-370) and fired_masks_update (:388-402): plain accessors of the reference
    surface."""
    import contextlib, io, types
    import torch
    from slak_amd.sparse_core import Masking
    net = torch.nn.Linear(4, 3)
    opt = torch.optim.Adam(net.parameters(), lr=0.1)
    net(torch.randn(2, 4)).sum().backward(); opt.step()
    args = types.SimpleNamespace(device="cpu", fix=False, update_frequency=None, only_L=False, sparse_init="uniform", sparsity=0.5, distributed=False)
    mk = Masking(opt, None, None, prune_mode="magnitude", growth_mode="gradient", redistribution_mode="none", args=args)
    st = opt.state[net.weight]
    assert torch.equal(mk.get_momentum_for_weight(net.weight), st["exp_avg"] / (torch.sqrt(st["exp_avg_sq"]) + 1e-08))
    mk.modules = [net]
    mk.masks = {"weight": torch.tensor([[1., 0, 0, 1], [0, 0, 1, 0], [0, 0, 0, 0]])}
    with contextlib.redirect_stdout(io.StringIO()):
        layer, total = mk.fired_masks_update()
        assert layer == {"weight": 3 / 12} and total == 3 / 12
        mk.masks["weight"] = torch.tensor([[0., 1, 0, 1], [0, 0, 0, 0], [0, 0, 0, 1]])
        layer, total = mk.fired_masks_update()
    assert total == 5 / 12                                # union of everything that was ever on
