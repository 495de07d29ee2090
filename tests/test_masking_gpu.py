"""GPU parity tests of the Masking prune / regrow / apply step (pytest -m gpu).

Mask index sets must be BIT-EXACT (BASELINE.json).  Checkers: (1) recorded runs of the UNMODIFIED reference
sparse_core.Masking on CPU (tests/golden/mask_*.npz), (2) the numpy oracle (oracle/mask_oracle.py, stable
tie policy) on tie-heavy data the reference itself is ambiguous on, (3) size-independent properties at the
full SLaK-T mask-set size (95 tensors / 30.7 M elements).
"""
import contextlib
import ctypes
import io
import math
import os
import sys
import types

import numpy as np
import pytest
import torch

import oracle
from conftest import GOLDEN, load_golden

pytestmark = pytest.mark.gpu
sys.path.insert(0, GOLDEN)


def _args(device, only_L=False, update_frequency=3, sparsity=0.4, distributed=False):
    return types.SimpleNamespace(device=str(device), fix=False, update_frequency=update_frequency, only_L=only_L,
                                 sparse_init="uniform", sparsity=sparsity, distributed=distributed)


# ------------------------------------------------------------------ (1) recorded reference runs
@pytest.mark.parametrize("tag,only_L,opt_kind", [("all_sgd", False, "sgd"), ("onlyL_adamw", True, "adamw"),
                                                 ("onlyL_adamw", True, "masked_adamw"),
                                                 ("all_adamw_random", False, "adamw"),           # growth = funcs.random_growth
                                                 ("all_adamw_random", False, "masked_adamw")])
def test_recorded_reference_run(tag, only_L, opt_kind, gpu):
    from make_golden import TinyNet                      # same tiny module the fixtures were recorded on
    from slak_amd.sparse_core import CosineDecay, Masking
    g = load_golden("mask_" + tag)
    nsteps, ufreq, T_max = (int(v) for v in g["meta"])
    torch.manual_seed(123)
    model = TinyNet()
    for p in model.parameters():
        p.data = torch.randn_like(p) * 0.05
    model = model.to(gpu)
    if opt_kind == "sgd":
        opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9)
    elif opt_kind == "masked_adamw":                      # AdamW + w *= mask as one HIP launch
        from slak_amd.optim_factory import MaskedAdamW
        opt = MaskedAdamW(model.parameters(), lr=1e-2, weight_decay=0.05)
    else:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-2, weight_decay=0.05)
    decay = CosineDecay(0.3, T_max)
    torch.manual_seed(7)                                 # same CPU RNG stream for the uniform init
    with contextlib.redirect_stdout(io.StringIO()):
        mask = Masking(opt, train_loader=None, prune_rate_decay=decay, prune_rate=0.3, prune_mode="magnitude",
                       growth_mode="random" if tag.endswith("_random") else "gradient", redistribution_mode="none",
                       args=_args(gpu, only_L))
        mask.add_module(model)
    names = [str(n) for n in g["names"]]
    assert list(mask.masks.keys()) == names
    params = dict(model.named_parameters())
    for n in names:
        np.testing.assert_array_equal(mask.masks[n].cpu().numpy(), g[f"m_init/{n}"], err_msg="init mask " + n)
    for n, p in params.items():
        np.testing.assert_array_equal(p.detach().cpu().numpy(), g[f"w_init/{n}"], err_msg="init weight " + n)
    for step in range(1, nsteps + 1):
        for n, p in params.items():
            p.grad = torch.from_numpy(g[f"g{step}/{n}"]).to(gpu)
        with contextlib.redirect_stdout(io.StringIO()):
            mask.step()
        assert mask.prune_rate == float(g["prune_rates"][step - 1])
        for n in names:                                   # bit-exact index sets
            np.testing.assert_array_equal(mask.masks[n].cpu().numpy(), g[f"m{step}/{n}"], err_msg=f"mask {n} step {step}")
        for n, p in params.items():                       # optimizer arithmetic is torch's (GPU vs CPU rounding)
            np.testing.assert_allclose(p.detach().cpu().numpy(), g[f"w{step}/{n}"], rtol=2e-5, atol=1e-7, err_msg=f"weight {n} step {step}")
        if step % ufreq == 0:
            for n in names:
                nz, zeros, removed = g[f"stats{step}/{n}"]
                assert (mask.name2nonzeros[n], mask.name2zeros[n], mask.name2removed[n]) == (nz, zeros, removed)
        opt.zero_grad(set_to_none=False)


# ------------------------------------------------------------------ helpers on the raw C ABI
class Plan:
    def __init__(self, weights, masks, grads=None, momenta=None):
        from slak_amd import _lib
        self._lib, self.L = _lib, _lib.lib()
        n = len(weights)
        segs = (_lib.MaskSegment * n)()
        for i in range(n):
            segs[i].weight, segs[i].mask = weights[i].data_ptr(), masks[i].data_ptr()
            segs[i].grad = grads[i].data_ptr() if grads is not None else None
            segs[i].momentum = momenta[i].data_ptr() if momenta is not None and momenta[i] is not None else None
            segs[i].numel = weights[i].numel()
        self.plan = ctypes.c_void_p()
        _lib.check(self.L.slak_mask_plan_create(segs, n, ctypes.byref(self.plan)), "create")
        self.n = n
        self.stream = torch.cuda.current_stream().cuda_stream

    def apply(self):
        self._lib.check(self.L.slak_mask_apply(self.plan, self.stream), "apply")

    def prune_and_grow(self, rate):
        self._lib.check(self.L.slak_mask_prune_and_grow(self.plan, float(rate), self.stream), "prune_and_grow")
        st = (ctypes.c_double * (4 * self.n))()
        self._lib.check(self.L.slak_mask_read_stats(self.plan, st, self.stream), "stats")
        return np.array(st).reshape(self.n, 4)

    def checksum(self):
        out = ctypes.c_ulonglong(0)
        self._lib.check(self.L.slak_mask_checksum(self.plan, ctypes.byref(out), self.stream), "checksum")
        return out.value

    def __del__(self):
        self.L.slak_mask_plan_destroy(self.plan)


def _oracle_step(ws, ms, gs, rate):
    names = [str(i) for i in range(len(ws))]
    nw, nm, st = oracle.truncate_weights(dict(zip(names, ws)), dict(zip(names, ms)), dict(zip(names, gs)), rate)
    return [nw[n] for n in names], [nm[n] for n in names], [st[n] for n in names]


def _run_case(shapes, rate, gpu, seed, quant=None, density=0.6, zero_grads=False):
    rng = np.random.default_rng(seed)
    ws, ms, gs = [], [], []
    for shp in shapes:
        w = rng.standard_normal(shp).astype(np.float32) * 0.05
        g = rng.standard_normal(shp).astype(np.float32)
        if quant:                                           # few distinct magnitudes -> massive ties at the cut
            w = (np.round(w * quant) / quant).astype(np.float32)
            g = (np.round(g * quant / 8) / (quant / 8)).astype(np.float32)
        if zero_grads:
            g[...] = 0
        m = (rng.random(shp) < density).astype(np.float32)
        ws.append((w * m).astype(np.float32)); ms.append(m); gs.append(g)
    tw = [torch.from_numpy(a.copy()).to(gpu) for a in ws]
    tm = [torch.from_numpy(a.copy()).to(gpu) for a in ms]
    tg = [torch.from_numpy(a.copy()).to(gpu) for a in gs]
    plan = Plan(tw, tm, tg)
    stats = plan.prune_and_grow(rate)
    torch.cuda.synchronize()
    ow, om, ost = _oracle_step(ws, ms, gs, rate)
    for i in range(len(shapes)):
        np.testing.assert_array_equal(tm[i].cpu().numpy(), om[i], err_msg=f"mask of tensor {i} shape {shapes[i]}")
        np.testing.assert_array_equal(tw[i].cpu().numpy(), ow[i], err_msg=f"weight of tensor {i}")
        assert stats[i, 0] == ost[i]["nonzeros"] and stats[i, 1] == ost[i]["zeros"] and stats[i, 2] == ost[i]["removed"]
        assert stats[i, 3] == om[i].sum()
    return plan


SHAPES = [(96, 1, 51, 5), (96, 1, 5, 51), (96, 1, 5, 5), (384, 96), (7,), (1, 1, 1, 1), (2049,), (4096, 75)]


@pytest.mark.parametrize("rate", [0.3, 0.005, 0.9999, 0.5])
def test_prune_grow_vs_oracle_continuous(rate, gpu):
    _run_case(SHAPES, rate, gpu, seed=1)


@pytest.mark.parametrize("quant", [2, 16, 200])
def test_prune_grow_vs_oracle_ties(quant, gpu):
    """Heavily tied keys: the index set is defined by the stable (lowest-index-first) policy."""
    _run_case(SHAPES, 0.37, gpu, seed=2, quant=quant)


def test_prune_grow_edge_rates_and_masks(gpu):
    _run_case(SHAPES, 0.0, gpu, seed=3)                       # num_remove == 0 -> mask = (w != 0)   (funcs.py:110)
    _run_case(SHAPES, 1.0, gpu, seed=4)                       # prune everything, regrow by gradient
    _run_case(SHAPES, 0.3, gpu, seed=5, density=1.1)          # all-ones masks
    _run_case(SHAPES, 0.3, gpu, seed=6, density=-1.0)         # all-zero masks
    _run_case(SHAPES, 0.3, gpu, seed=7, zero_grads=True)      # all growth keys tie at 0 -> lowest indices


def test_prune_grow_block_boundaries_unaligned_views_and_unreachable_taps(gpu):
    """The five-pass select: tensors around the 2048 / 16384-element block sizes, bases that are not 16-byte aligned (views into a
    flat bucket take the scalar path), and gradients that are exactly zero on most inactive weights -- the taps of a 51 x 5
    kernel a 7 x 7 plane never reaches -- so that the regrow cut falls inside the ties at key 0."""
    shapes = [(16383,), (16384,), (16385,), (32768 + 3,), (3 * 16384 + 2047,), (768, 1, 51, 5)]
    rng = np.random.default_rng(21)
    ws, ms, gs = [], [], []
    for shp in shapes:
        w = rng.standard_normal(shp).astype(np.float32) * 0.05
        g = rng.standard_normal(shp).astype(np.float32)
        if len(shp) == 4:
            g[:, :, :19, :] = 0; g[:, :, 32:, :] = 0             # rows a 7-row plane cannot reach
        m = (rng.random(shp) < 0.6).astype(np.float32)
        ws.append((w * m).astype(np.float32)); ms.append(m); gs.append(g)
    for shift in (0, 1, 3):                                     # element offset of every tensor inside its flat buffer
        def put(arrs):
            out = []
            for a in arrs:
                flat = torch.zeros(a.size + 8, device=gpu)
                v = flat[shift:shift + a.size].view(a.shape)
                v.copy_(torch.from_numpy(a))
                out.append(v)
            return out
        tw, tm, tg = put(ws), put(ms), put(gs)
        tmom = put([np.ones_like(a) for a in ws])
        plan = Plan(tw, tm, tg, momenta=tmom)
        stats = plan.prune_and_grow(0.3)
        ow, om, ost = _oracle_step(ws, ms, gs, 0.3)
        for i in range(len(shapes)):
            np.testing.assert_array_equal(tm[i].cpu().numpy(), om[i], err_msg=f"mask {shapes[i]} shift {shift}")
            np.testing.assert_array_equal(tw[i].cpu().numpy(), ow[i], err_msg=f"weight {shapes[i]} shift {shift}")
            np.testing.assert_array_equal(tmom[i].cpu().numpy(), om[i], err_msg=f"momentum {shapes[i]} shift {shift}")
            assert stats[i, 2] == ost[i]["removed"] and stats[i, 3] == om[i].sum()
    # the regrow cut of the last tensor is inside the zero-gradient ties: fewer weights come back than were removed
    assert stats[-1, 3] < stats[-1, 0]


def test_apply_mask_and_momentum(gpu):
    torch.manual_seed(0)
    w = torch.randn(5000, device=gpu); m = (torch.rand(5000, device=gpu) < 0.5).float(); mom = torch.randn(5000, device=gpu)
    w2 = torch.randn(3, 1, 7, 5, device=gpu); m2 = (torch.rand(3, 1, 7, 5, device=gpu) < 0.5).float()
    ew, emom, ew2 = (w * m).clone(), (mom * m).clone(), (w2 * m2).clone()
    plan = Plan([w, w2], [m, m2], momenta=[mom, None])
    plan.apply(); plan.apply()                                  # idempotent
    assert torch.equal(w, ew) and torch.equal(mom, emom) and torch.equal(w2, ew2)
    c1 = plan.checksum()
    m2[0, 0, 0, 0] = 1 - m2[0, 0, 0, 0]
    assert plan.checksum() != c1


def test_large_segment_vs_torch_stable_sort(gpu):
    """One 4.19 M-element tensor (SLaK-B pwconv, the largest in any config) against torch.sort(stable=True)."""
    torch.manual_seed(11)
    n = 1024 * 4096
    w = torch.randn(n, device=gpu) * 0.02
    w = (w * 4096).round() / 4096                               # ties
    m = (torch.rand(n, device=gpu) < 0.6).float()
    w = w * m
    g = torch.randn(n, device=gpu)
    rate = 0.21
    nz = m.sum().item(); zeros = n - nz
    k = math.ceil(zeros + math.ceil(rate * nz))
    idx = torch.sort(w.abs(), stable=True)[1]
    em = m.clone(); em[idx[:k]] = 0
    removed = int(nz - em.sum().item())
    gi = torch.sort((g * (em == 0).float()).abs(), descending=True, stable=True)[1]
    em[gi[:removed]] = 1
    plan = Plan([w], [m], [g])
    st = plan.prune_and_grow(rate)
    assert torch.equal(m, em)
    assert st[0, 2] == removed and st[0, 3] == em.sum().item()


def test_full_slak_t_mask_set_properties(gpu):
    """The SLaK-T k51 mask set: 95 tensors / 30.7 M elements.  Properties that hold for any input:
    masks are 0/1, per-tensor nonzero count is conserved by prune+regrow (continuous data), dead weights are
    exactly zero, the step is deterministic (checksum), apply is idempotent."""
    from slak_amd.slak_model import slak_mask_set_shapes
    shapes = slak_mask_set_shapes("tiny")
    assert len(shapes) == 95 and sum(int(np.prod(s)) for s in shapes) == 30717984
    torch.manual_seed(5)
    ws = [torch.randn(s, device=gpu) * 0.02 for s in shapes]
    ms = [(torch.rand(s, device=gpu) < 0.6).float() for s in shapes]
    gs = [torch.randn(s, device=gpu) for s in shapes]
    for w, m in zip(ws, ms):
        w.mul_(m)
    ws0 = [w.clone() for w in ws]; ms0 = [m.clone() for m in ms]
    plan = Plan(ws, ms, gs)
    before = [m.sum().item() for m in ms]
    st = plan.prune_and_grow(0.3)
    c1 = plan.checksum()
    for i, (w, m) in enumerate(zip(ws, ms)):
        assert torch.all((m == 0) | (m == 1))
        assert m.sum().item() == before[i] == st[i, 0] == st[i, 3]
        assert st[i, 2] == math.ceil(0.3 * before[i])
        assert torch.all(w[m == 0] == 0)
        kept = (m == 1) & (ms0[i] == 1)
        assert torch.equal(w[kept], ws0[i][kept])
    # determinism: same inputs -> same masks
    ws2 = [w.clone() for w in ws0]; ms2 = [m.clone() for m in ms0]
    plan2 = Plan(ws2, ms2, gs)
    plan2.prune_and_grow(0.3)
    assert plan2.checksum() == c1
    for a, b in zip(ms, ms2):
        assert torch.equal(a, b)
