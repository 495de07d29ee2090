"""numpy restatement of the reference's dynamic-sparsity step (TEST INFRASTRUCTURE ONLY).

Follows, line by line:
  * ``magnitude_prune``    funcs.py:107-114
  * ``gradient_growth``    funcs.py:196-205  (+ ``get_gradient_for_weights`` sparse_core.py:372-374)
  * ``Masking.apply_mask`` sparse_core.py:316-333 (fp32 branch, ``tensor.data * mask``)
  * ``Masking.truncate_weights`` sparse_core.py:335-357 (prune loop, then growth loop, then apply)
  * ``CosineDecay``        sparse_core.py:49-64 (closed form of torch CosineAnnealingLR)

Tie policy: the reference calls ``torch.sort`` without ``stable=True``, so its
result is arbitrary among equal keys.  This oracle -- and the HIP kernels -- define ties by LOWEST
FLAT INDEX FIRST, i.e. ``torch.sort(..., stable=True)`` / ``np.argsort(kind="stable")``.
Pinned by tests/test_oracle.py against tests/golden/mask_*.npz, produced by running the reference's
own ``sparse_core.Masking`` + ``funcs`` (imported unmodified from /root/reference in the build
container) via tests/golden/make_golden.py.
"""
import math

import numpy as np


def magnitude_prune(mask, weight, prune_rate, nonzeros, zeros):
    """funcs.py:107-114.  ``mask`` float32 array (modified copy returned); ``nonzeros``/``zeros`` are the
    float counts ``Masking.truncate_weights`` stored in name2nonzeros/name2zeros (sparse_core.py:341-342)."""
    num_remove = math.ceil(prune_rate * nonzeros)
    k = math.ceil(zeros + num_remove)
    if num_remove == 0.0:
        return (weight != 0.0).astype(np.float32)
    idx = np.argsort(np.abs(weight.reshape(-1)), kind="stable")
    out = mask.astype(np.float32).copy().reshape(-1)
    out[idx[:k]] = 0.0
    return out.reshape(mask.shape)


def gradient_growth(new_mask, total_regrowth, grad):
    """funcs.py:196-205.  ``new_mask`` uint8/float array; returns float32 mask."""
    g = grad.astype(np.float32) * (new_mask == 0).astype(np.float32)
    key = np.abs(g).reshape(-1)
    # descending stable sort == ascending stable sort of the negated key (ties keep index order)
    idx = np.argsort(-key, kind="stable")
    out = new_mask.astype(np.float32).copy().reshape(-1)
    if total_regrowth > 0:
        out[idx[:int(total_regrowth)]] = 1.0
    return out.reshape(new_mask.shape)


def random_growth(new_mask, total_regrowth, rand):
    """funcs.py:170-175.  ``rand``: the float32 numbers the reference draws with ``torch.rand(new_mask.shape)`` (host generator);
    the comparison against the Python float p happens in float32 (torch compares a float32 tensor with a Python scalar in the
    tensor's dtype)."""
    n = int((new_mask == 0).sum())
    if n == 0:
        return new_mask.astype(np.float32)
    p = np.float32(total_regrowth / n)
    return ((new_mask != 0) | (rand.astype(np.float32) < p)).astype(np.float32)


def apply_mask(weight, mask):
    """sparse_core.py:326  ``tensor.data = tensor.data*self.masks[name]`` (fp32)."""
    return (weight.astype(np.float32) * mask.astype(np.float32)).astype(np.float32)


def truncate_weights(weights, masks, grads, prune_rate, growth="gradient", rands=None):
    """sparse_core.py:335-357 over dicts name -> array.  Returns (new_weights, new_masks, stats).  growth: "gradient" (needs
    ``grads``) or "random" (needs ``rands``: name -> the numbers torch.rand produced for that tensor)."""
    names = list(masks.keys())
    new_masks, stats = {}, {}
    for name in names:                                   # prune loop  :337-347
        mask = masks[name]
        nonzeros = float(mask.sum(dtype=np.float64))
        zeros = mask.size - nonzeros
        nm = magnitude_prune(mask, weights[name], prune_rate, nonzeros, zeros)
        removed = nonzeros - float(nm.sum(dtype=np.float64))
        new_masks[name] = nm
        stats[name] = dict(nonzeros=nonzeros, zeros=zeros, removed=removed)
    for name in names:                                   # growth loop :349-355
        nm = new_masks[name].astype(np.uint8)            # ``.data.byte()``
        if growth == "random":
            new_masks[name] = random_growth(nm, math.floor(stats[name]["removed"]), rands[name])
        else:
            new_masks[name] = gradient_growth(nm, math.floor(stats[name]["removed"]), grads[name])
    new_weights = {n: (apply_mask(weights[n], new_masks[n]) if n in new_masks else weights[n]) for n in weights}
    return new_weights, new_masks, stats


def cosine_prune_rate(prune_rate0, T_max, step, eta_min=0.005):
    """Closed form of sparse_core.CosineDecay (sparse_core.py:49-64): lr of CosineAnnealingLR after
    ``step`` scheduler steps.  Used only as a cross-check; the product keeps the torch scheduler object
    on the host so the value is bit-identical to the reference's."""
    return eta_min + (prune_rate0 - eta_min) * (1 + math.cos(math.pi * step / T_max)) / 2
