"""GPU parity tests of the mask-aware optimizer step and the EMA update (pytest -m gpu).

EMA: bit-exact against recorded runs of the UNMODIFIED reference model_sema.ModelEma (tests/golden/ema_*.npz) and against the numpy
oracle (oracle/optim_ema_oracle.py) on ragged / unaligned / large entries.  AdamW: torch.optim.AdamW's arithmetic to within a few
float32 roundings per step (tolerance stated at each check; torch's own CPU and GPU kernels differ by as much), the mask
application exact.
"""
import contextlib
import ctypes
import io
import sys
import types

import numpy as np
import pytest
import torch

import oracle
from oracle import optim_ema_oracle as O
from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
sys.path.insert(0, GOLDEN)


def _bits(a):
    return a.view(np.int32) if a.dtype == np.float32 else a


# ------------------------------------------------------------------ EMA
@pytest.mark.parametrize("tag", ["masked", "dense"])
def test_ema_recorded_reference_run(tag, gpu):
    from make_golden import TinyNet
    from slak_amd.model_sema import ModelEma
    g = load_golden("ema_" + tag)
    keys = [str(k) for k in g["keys"]]
    mask_names = [str(n) for n in g["mask_names"]]
    model = TinyNet().to(gpu)
    ema = ModelEma(model, decay=float(g["decay"]))
    assert list(ema.ema.state_dict().keys()) == keys
    ema.ema.load_state_dict({k: torch.from_numpy(g[f"e0/{k}"]) for k in keys})
    fake = types.SimpleNamespace(masks={n: torch.zeros_like(dict(model.named_parameters())[n]) for n in mask_names}) if mask_names else None
    for step in range(1, int(g["meta"][0]) + 1):
        model.load_state_dict({k: torch.from_numpy(g[f"w{step}/{k}"]) for k in keys})      # in place: pointers stay
        for n in mask_names:
            fake.masks[n].copy_(torch.from_numpy(g[f"m{step}/{n}"]))
        ema.update(model, fake)
        got = ema.ema.state_dict()
        for k in keys:
            np.testing.assert_array_equal(_bits(got[k].cpu().numpy()), _bits(g[f"e{step}/{k}"]), err_msg=f"{k} step {step}")


def test_ema_ragged_entries_vs_oracle(gpu):
    """C ABI directly: unaligned views, sizes around the 4096-element chunk, a 3-element and a 1-element entry, an int64 counter,
    a 5 M-element entry; masked and dense."""
    from slak_amd import _lib
    L = _lib.lib()
    gen = torch.Generator().manual_seed(5)
    sizes = [1, 3, 4095, 4096, 4097, 12288, 70001, 5_000_003]
    ents = []
    for i, n in enumerate(sizes):
        off = i % 3                                            # views starting 0, 4 or 8 bytes into an allocation
        e = (torch.randn(n + 3, generator=gen) * 0.1)
        w = torch.randn(n + 3, generator=gen) * 0.1
        m = (torch.rand(n + 3, generator=gen) < 0.5).float() if i % 2 == 0 else None
        if m is not None:
            e = e * (torch.rand(n + 3, generator=gen) < 0.7).float()          # some entries of the EMA are still zero under mask == 1
        ents.append([t if t is None else t.to(gpu)[off:off + n] for t in (e, w, m)])
    cnt_e = torch.tensor([7, 123456789], dtype=torch.int64, device=gpu)
    cnt_w = torch.tensor([9, 123456999], dtype=torch.int64, device=gpu)
    segs = (_lib.EmaSegment * (len(ents) + 1))()
    for i, (e, w, m) in enumerate(ents):
        segs[i].ema, segs[i].model, segs[i].mask = e.data_ptr(), w.data_ptr(), (m.data_ptr() if m is not None else None)
        segs[i].numel, segs[i].dtype = e.numel(), _lib.SLAK_F32
    segs[len(ents)].ema, segs[len(ents)].model, segs[len(ents)].mask = cnt_e.data_ptr(), cnt_w.data_ptr(), None
    segs[len(ents)].numel, segs[len(ents)].dtype = 2, 3
    want = [O.ema_update_entry(e.cpu().numpy(), w.cpu().numpy(), 0.9999, None if m is None else m.cpu().numpy()) for e, w, m in ents]
    want_cnt = O.ema_update_entry(cnt_e.cpu().numpy(), cnt_w.cpu().numpy(), 0.9999)
    h = ctypes.c_void_p()
    _lib.check(L.slak_ema_plan_create(segs, len(ents) + 1, ctypes.byref(h)), "create")
    _lib.check(L.slak_ema_update(h, 0.9999, torch.cuda.current_stream().cuda_stream), "update")
    torch.cuda.synchronize()
    for (e, _, _), ref in zip(ents, want):
        np.testing.assert_array_equal(_bits(e.cpu().numpy()), _bits(ref))
    np.testing.assert_array_equal(cnt_e.cpu().numpy(), want_cnt)
    _lib.check(L.slak_ema_plan_destroy(h), "destroy")


# ------------------------------------------------------------------ AdamW (+ mask, + bf16 copy)
def _clone_params(ps, device):
    return [torch.nn.Parameter(p.detach().clone().to(device)) for p in ps]


def test_masked_adamw_matches_torch_adamw(gpu):
    """Three groups with their own lr / weight decay / betas, ragged sizes, one parameter that never gets a gradient, 6 steps with a
    changing lr.  Tolerance: 3e-6 of each tensor's max magnitude (torch's foreach / fused / CPU AdamW differ among themselves at
    the 1e-7 level per step; errors accumulate through exp_avg)."""
    from slak_amd.optim_factory import MaskedAdamW
    gen = torch.Generator().manual_seed(0)
    shapes = [(96, 1, 51, 5), (384, 96), (96,), (5,), (1,), (33, 7, 3), (4097,), (70001,)]
    base = [torch.randn(s, generator=gen) * 0.05 for s in shapes]
    pa, pb = _clone_params(base, gpu), _clone_params(base, gpu)
    def groups(ps):
        return [dict(params=ps[:2], lr=4e-3, weight_decay=0.05), dict(params=ps[2:5], lr=1e-3, weight_decay=0.0, betas=(0.8, 0.99)),
                dict(params=ps[5:], lr=2e-3, weight_decay=0.3, eps=1e-6)]
    ours, ref = MaskedAdamW(groups(pa), lr=1e-3), torch.optim.AdamW(groups(pb), lr=1e-3, foreach=False)
    for step in range(6):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if i == 3:
                continue                                        # never receives a gradient
            if i == 6 and step % 2 == 1:
                a.grad = b.grad = None                          # skipped every other step: its own step count
                continue
            gr = (torch.randn(a.shape, generator=gen) * 0.1).to(gpu)
            a.grad, b.grad = gr.clone(), gr.clone()
        for o in (ours, ref):
            for g in o.param_groups:
                g["lr"] = g["lr"] * 0.9
        ours.step(); ref.step()
        for i, (a, b) in enumerate(zip(pa, pb)):
            tol = 3e-6 * float(b.detach().abs().max()) + 1e-12
            assert float((a.detach() - b.detach()).abs().max()) <= tol, (step, i)
    np.testing.assert_array_equal(pa[3].detach().cpu().numpy(), base[3].numpy())           # untouched
    sa, sb = ours.state_dict()["state"], ref.state_dict()["state"]
    assert float(sa[6]["step"]) == float(sb[6]["step"]) == 3.0 and float(sa[0]["step"]) == 6.0
    for i in (0, 1, 7):
        for k in ("exp_avg", "exp_avg_sq"):
            assert float((sa[i][k] - sb[i][k]).abs().max()) <= 3e-6 * float(sb[i][k].abs().max())


def test_masked_adamw_mask_and_bf16_copy(gpu):
    """With masks the result is EXACTLY (unmasked result) * mask; the bf16 copy is EXACTLY the rounded new weight and the cache of
    slak_amd.block_ops is left current (no refresh launch needed)."""
    from slak_amd import block_ops
    from slak_amd.optim_factory import MaskedAdamW
    gen = torch.Generator().manual_seed(1)
    base = [torch.randn(s, generator=gen) * 0.05 for s in [(384, 96), (96, 384), (4099,)]]
    pa, pb = _clone_params(base, gpu), _clone_params(base, gpu)
    masks = {pa[0]: (torch.rand(base[0].shape, generator=gen) < 0.6).float().to(gpu), pa[2]: (torch.rand(base[2].shape, generator=gen) < 0.3).float().to(gpu)}
    old = block_ops.cache_lowp_weights
    block_ops.cache_lowp_weights = True
    try:
        c0 = block_ops.lowp_param(pa[0]); c1 = block_ops.lowp_param(pa[1])
        ours, plain = MaskedAdamW(pa, lr=1e-2, weight_decay=0.05), MaskedAdamW(pb, lr=1e-2, weight_decay=0.05)
        ours.set_masks(masks)
        for step in range(3):
            for a, b in zip(pa, pb):
                gr = (torch.randn(a.shape, generator=gen) * 0.1).to(gpu)
                a.grad, b.grad = gr.clone(), gr.clone()
            with torch.no_grad():                               # keep both trajectories on the masked weights
                for a, b in zip(pa, pb):
                    b.copy_(a)
                for k in ("exp_avg", "exp_avg_sq"):
                    for a, b in zip(pa, pb):
                        if k in ours.state[a]:
                            plain.state[b][k] = ours.state[a][k].clone(); plain._state_dirty = True
            ours.step(); plain.step()
            for a, b in zip(pa, pb):
                want = b.detach() * masks[a] if a in masks else b.detach()
                assert torch.equal(a.detach(), want)
            for p, c in ((pa[0], c0), (pa[1], c1)):
                assert block_ops.lowp_param(p) is c                       # same buffer, already current
                assert torch.equal(c, p.detach().to(torch.bfloat16))
    finally:
        block_ops.cache_lowp_weights = old
        block_ops._lowp_cache.clear()


def test_masked_adamw_state_dict_interchanges_with_torch(gpu):
    from slak_amd.optim_factory import MaskedAdamW
    gen = torch.Generator().manual_seed(2)
    base = [torch.randn(s, generator=gen) * 0.05 for s in [(10, 7), (7,)]]
    pa, pb, pc = _clone_params(base, gpu), _clone_params(base, gpu), _clone_params(base, gpu)
    ours, ref = MaskedAdamW(pa, lr=1e-2), torch.optim.AdamW(pb, lr=1e-2, foreach=False)
    grads = [[(torch.randn(p.shape, generator=gen) * 0.1).to(gpu) for p in pa] for _ in range(4)]
    for s in range(2):
        for a, b, g in zip(pa, pb, grads[s]):
            a.grad, b.grad = g.clone(), g.clone()
        ours.step(); ref.step()
    # torch -> ours: continue the torch run with MaskedAdamW; ours -> torch the other way round
    cont_ours = MaskedAdamW(pc, lr=1e-2)
    with torch.no_grad():
        for c, b in zip(pc, pb):
            c.copy_(b)
    import copy                                                  # a checkpoint file round trip copies; torch's load_state_dict itself aliases tensors
    cont_ours.load_state_dict(copy.deepcopy(ref.state_dict()))
    ref2 = torch.optim.AdamW(pa, lr=1e-2, foreach=False)
    ref2.load_state_dict(copy.deepcopy(ours.state_dict()))
    for s in range(2, 4):
        for a, b, c, g in zip(pa, pb, pc, grads[s]):
            a.grad, b.grad, c.grad = g.clone(), g.clone(), g.clone()
        ref.step(); cont_ours.step(); ref2.step()
    for a, b, c in zip(pa, pb, pc):
        tol = 3e-6 * float(b.detach().abs().max())
        assert float((c.detach() - b.detach()).abs().max()) <= tol
        assert float((a.detach() - b.detach()).abs().max()) <= tol
