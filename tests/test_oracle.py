"""The oracle is pinned before it is trusted (CPU, no GPU):
 * C restatement (oracle/dwconv_oracle.c) vs the committed fixtures produced by the reference's own
   ground truth F.conv2d(+autograd)  (test_correctness.py:8-9, :67-90)
 * numpy mask restatement (oracle/mask_oracle.py) vs fixtures produced by the UNMODIFIED reference
   sparse_core.Masking / funcs running on CPU (tests/golden/make_golden.py).
"""
import math

import numpy as np
import pytest

import oracle
from conftest import golden_conv_cases, load_golden


@pytest.mark.parametrize("case", golden_conv_cases())
def test_c_oracle_matches_reference_conv(case):
    g = load_golden("dwconv_" + case)
    N, C, H, W, kh, kw, _ = g["meta"]
    y = oracle.dwconv2d_fwd(g["x"], g["w"])
    dx = oracle.dwconv2d_bwd_data(g["dy"], g["w"])
    dw = oracle.dwconv2d_bwd_filter(g["dy"], g["x"], int(kh), int(kw))
    # both sides accumulate in fp64: agreement is at fp64 round-off of a <=961-term sum
    np.testing.assert_allclose(y, g["y64"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(dx, g["dx64"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(dw, g["dw64"], rtol=1e-11, atol=1e-11)


def test_c_oracle_properties():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 10, 9)).astype(np.float32)
    w = rng.standard_normal((3, 1, 7, 5)).astype(np.float32)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    # adjointness: <conv(x,w), dy> == <x, dgrad(dy,w)> == <w, wgrad(dy,x)>
    a = (oracle.dwconv2d_fwd(x, w) * dy).sum()
    b = (oracle.dwconv2d_bwd_data(dy, w) * x).sum()
    c = (oracle.dwconv2d_bwd_filter(dy, x, 7, 5) * w).sum()
    assert abs(a - b) < 1e-9 * abs(a) + 1e-9 and abs(a - c) < 1e-9 * abs(a) + 1e-9
    # identity kernel
    wi = np.zeros((3, 1, 7, 5), np.float32); wi[:, 0, 3, 2] = 1
    np.testing.assert_array_equal(oracle.dwconv2d_fwd(x, wi), x.astype(np.float64))


def test_bf16_round_matches_torch():
    import torch
    v = torch.randn(10000) * 3
    v[:4] = torch.tensor([0.0, -0.0, 1.00390625, 65504.0])
    ref = v.to(torch.bfloat16).float().numpy()
    np.testing.assert_array_equal(oracle.bf16_round(v.numpy()), ref)


def test_mask_funcs_match_reference():
    g = load_golden("mask_funcs")
    m, w = g["m"], g["w"]
    nz = float(m.sum()); zeros = m.size - nz
    pruned = oracle.magnitude_prune(m, w, float(g["rate"]), nz, zeros)
    np.testing.assert_array_equal(pruned, g["pruned"])
    removed = nz - float(pruned.sum())
    assert removed == float(g["removed"])
    grown = oracle.gradient_growth(pruned.astype(np.uint8), math.floor(removed), g["grad"])
    np.testing.assert_array_equal(grown, g["grown"])


@pytest.mark.parametrize("tag", ["all_sgd", "onlyL_adamw"])
def test_mask_truncate_matches_reference_masking(tag):
    """At every update step of the recorded run, oracle.truncate_weights reproduces the reference's
    masks bit-exactly.  State just before truncate_weights (sparse_core.py:309-311) is reconstructed
    from the recording: weights there are  w_after_step  on positions that survive, and the prune
    decision only looks at |w| of the (optimizer-stepped, mask-applied) weights."""
    g = load_golden("mask_" + tag)
    nsteps, ufreq, T_max = (int(v) for v in g["meta"])
    names = [str(n) for n in g["names"]]
    for step in range(ufreq, nsteps + 1, ufreq):
        m_prev = {n: g[f"m{step-1}/{n}" if step > 1 else f"m_init/{n}"] for n in names}
        grads = {n: g[f"g{step}/{n}"] for n in names}
        w_pre = {n: g[f"wpre{step}/{n}"] for n in names}
        rate = float(g[f"rate_at{step}"])
        assert rate == float(g["prune_rates"][step - 1])
        w_new, m_new, stats = oracle.truncate_weights(w_pre, m_prev, grads, rate)
        for n in names:
            np.testing.assert_array_equal(m_new[n], g[f"m{step}/{n}"], err_msg=f"{n} step {step}")
            np.testing.assert_array_equal(w_new[n], g[f"w{step}/{n}"], err_msg=f"{n} step {step}")
            nz, zeros, removed = g[f"stats{step}/{n}"]
            assert (stats[n]["nonzeros"], stats[n]["zeros"], stats[n]["removed"]) == (nz, zeros, removed)
    # ordinary steps: apply_mask only (sparse_core.py:302, :326)
    for s_ in range(1, nsteps + 1):
        for n in names:
            assert np.all(g[f"w{s_}/{n}"][g[f"m{s_}/{n}"] == 0] == 0)
    # cosine schedule closed form == recorded torch scheduler values
    for s_ in range(1, nsteps + 1):
        assert abs(oracle.cosine_prune_rate(0.3, T_max, s_) - g["prune_rates"][s_ - 1]) < 1e-12


# ------------------------------------------------------------------ EMA update and the AdamW step
def _bits(a):
    return a.view(np.int32) if a.dtype == np.float32 else a


@pytest.mark.parametrize("tag", ["masked", "dense"])
def test_ema_oracle_matches_reference_model_ema(tag):
    """oracle/optim_ema_oracle.py vs the recorded run of the UNMODIFIED reference ModelEma: bit-exact over 7 updates, including
    the entries regrown by prune-and-grow and BatchNorm's int64 counters."""
    from oracle import optim_ema_oracle as O
    g = load_golden("ema_" + tag)
    keys = [str(k) for k in g["keys"]]
    names = [str(n) for n in g["mask_names"]]
    e = {k: g[f"e0/{k}"] for k in keys}
    regrown = 0
    for step in range(1, int(g["meta"][0]) + 1):
        masks = {n: g[f"m{step}/{n}"] for n in names}
        regrown += sum(int(((e[n] == 0) & (masks[n] != 0)).sum()) for n in names)
        e = O.ema_update(e, {k: g[f"w{step}/{k}"] for k in keys}, float(g["decay"]), masks)
        for k in keys:
            np.testing.assert_array_equal(_bits(e[k]), _bits(g[f"e{step}/{k}"]), err_msg=f"{k} step {step}")
    assert (regrown > 0) == (tag == "masked")
    assert any(g[f"e0/{k}"].dtype == np.int64 for k in keys)


def test_adamw_oracle_matches_torch_adamw_with_reference_masking():
    """oracle adamw_step (+ mask) vs the recorded torch.optim.AdamW + reference Masking run on CPU.  torch's CPU kernels contract
    some of the multiply-adds, so agreement is to float32 round-off per step: 1e-6 of each tensor's magnitude over 7 steps."""
    from oracle import optim_ema_oracle as O
    g = load_golden("mask_onlyL_adamw")
    pnames, mnames = [str(n) for n in g["param_names"]], [str(n) for n in g["names"]]
    nsteps, ufreq, _ = (int(v) for v in g["meta"])
    w = {n: g[f"w_init/{n}"] for n in pnames}
    m = {n: np.zeros_like(w[n]) for n in pnames}
    v = {n: np.zeros_like(w[n]) for n in pnames}
    for step in range(1, nsteps + 1):
        for n in pnames:
            mask = (g[f"m{step - 1}/{n}"] if step > 1 else g[f"m_init/{n}"]) if n in mnames else None
            w[n], m[n], v[n] = O.adamw_step(w[n], g[f"g{step}/{n}"], m[n], v[n], step, 1e-2, 0.9, 0.999, 1e-8, 0.05, mask)
            if n in mnames and step % ufreq == 0:
                w[n] = w[n] * g[f"m{step}/{n}"]                      # truncate_weights ends with apply_mask (sparse_core.py:357)
            ref = g[f"w{step}/{n}"]
            assert np.abs(ref - w[n]).max() <= 1e-6 * np.abs(ref).max() + 1e-12, (n, step)


def test_mask_truncate_random_growth_matches_reference_masking():
    """funcs.random_growth (the default growth of main.py:211) on the recorded reference run: the random numbers are the host
    generator's, so replaying its stream (seed 7: one draw per masked tensor at init, one per masked tensor per prune-and-grow
    round, in named_parameters order) reproduces the reference's masks bit-exactly."""
    import torch
    g = load_golden("mask_all_adamw_random")
    nsteps, ufreq, _ = (int(v) for v in g["meta"])
    names = [str(n) for n in g["names"]]
    torch.manual_seed(7)
    for n in names:                                                   # uniform init, sparse_core.py:176-183
        r = torch.rand(g[f"m_init/{n}"].shape)
        np.testing.assert_array_equal((r < 0.6).float().numpy(), g[f"m_init/{n}"], err_msg="replay of the init draws: " + n)
    grown_total = 0
    for step in range(ufreq, nsteps + 1, ufreq):
        m_prev = {n: g[f"m{step-1}/{n}"] for n in names}
        w_pre = {n: g[f"wpre{step}/{n}"] for n in names}
        rands = {n: torch.rand(m_prev[n].shape).numpy() for n in names}
        w_new, m_new, stats = oracle.truncate_weights(w_pre, m_prev, None, float(g[f"rate_at{step}"]), growth="random", rands=rands)
        for n in names:
            np.testing.assert_array_equal(m_new[n], g[f"m{step}/{n}"], err_msg=f"{n} step {step}")
            np.testing.assert_array_equal(w_new[n], g[f"w{step}/{n}"], err_msg=f"{n} step {step}")
            nz, zeros, removed = g[f"stats{step}/{n}"]
            assert (stats[n]["nonzeros"], stats[n]["zeros"], stats[n]["removed"]) == (nz, zeros, removed)
            grown_total += int(((m_new[n] != 0) & (m_prev[n] == 0)).sum())
    assert grown_total > 0


def test_snip_layerwise_sparsities_match_reference():
    """slak_amd.sparse_core.SNIP (one-time host logic of --sparse_init snip, the README recipes' init) against the sparsities the
    reference's SNIP (sparse_core.py:11-47) produced for the same net and batch (tests/golden/make_golden.py --only snip)."""
    import sys, types
    import torch
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_golden import _snip_net
    from slak_amd.sparse_core import SNIP
    g = load_golden("snip_small")
    net = _snip_net()
    before = [p.detach().clone() for p in net.parameters()]
    masks = {str(n): None for n in g["names"]}
    sp = SNIP(net, float(g["keep_ratio"]), [(torch.from_numpy(g["images"]), torch.from_numpy(g["labels"]))], torch.device("cpu"), masks,
              types.SimpleNamespace(distributed=False))
    assert list(sp) == list(g["sparsities"])
    assert all(torch.equal(a, b) for a, b in zip(before, net.parameters())) and all(p.grad is None for p in net.parameters())


@pytest.mark.parametrize("tag", ["s40", "s90"])
def test_erk_init_matches_reference(tag):
    """Masking.init(mode='ERK') (sparse_core.py:184-262): same epsilon iteration, same per-layer densities, same host-generator draws,
    same (almost) dense layers popped -- masks identical to the reference's for the same seed.  The device apply at the end of
    init() is stubbed: this pins the one-time host logic."""
    import contextlib, io, sys, types
    import torch
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    from make_golden import TinyNet
    from slak_amd.sparse_core import CosineDecay, Masking
    g = load_golden("erk_init")
    torch.manual_seed(123)
    model = TinyNet()
    args = types.SimpleNamespace(device="cpu", fix=False, update_frequency=3, only_L=False, sparse_init="ERK",
                                 sparsity=float(g[f"{tag}/sparsity"]), distributed=False)
    torch.manual_seed(9)
    with contextlib.redirect_stdout(io.StringIO()):
        mk = Masking(torch.optim.SGD(model.parameters(), lr=0.1), None, CosineDecay(0.3, 20), prune_rate=0.3, prune_mode="magnitude",
                     growth_mode="gradient", redistribution_mode="none", args=args)
        mk.apply_mask = lambda: None                               # (device kernel; not part of what is pinned here)
        mk.add_module(model)
    names = [str(n) for n in g[f"{tag}/names"]]
    assert list(mk.masks.keys()) == names
    for n in names:
        np.testing.assert_array_equal(mk.masks[n].numpy(), g[f"{tag}/m/{n}"], err_msg=n)


def test_reference_step_loop_loads_from_source_and_from_bytecode():
    """oracle/ref_modules.load_engine: engine.py + utils.py of the reference, unmodified, importable in this image with only their
    import boundary supplied -- from the checkout here, and from the bytecode under oracle/_ref/ (the path the GPU box takes)."""
    import inspect
    import os
    import sys
    from oracle import ref_modules
    if not os.path.isdir(ref_modules.REF) and not ref_modules.available("reference_engine"):
        pytest.skip("neither /root/reference nor oracle/_ref/ is present")
    ref_modules.build()
    want = ["model", "criterion", "data_loader", "optimizer", "device", "epoch", "loss_scaler", "max_norm", "model_ema", "mixup_fn", "log_writer",
            "wandb_logger", "start_steps", "lr_schedule_values", "wd_schedule_values", "num_training_steps_per_epoch", "update_freq", "use_amp", "mask"]
    saved = ref_modules.FORCE_BYTECODE
    try:
        for force in (False, True):
            ref_modules.FORCE_BYTECODE = force
            engine, utils = ref_modules.load_engine()
            assert list(inspect.signature(engine.train_one_epoch).parameters) == want       # engine.py:17-22
            assert engine.utils is utils and sys.modules.get("utils") is not utils          # engine.py:15 got the reference's utils.py; nothing stays behind
            sched = utils.cosine_scheduler(4e-3, 1e-6, 2, 5, warmup_epochs=0)
            assert len(sched) == 10 and sched[0] == 4e-3
    finally:
        ref_modules.FORCE_BYTECODE = saved
