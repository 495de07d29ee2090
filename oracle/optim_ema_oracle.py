"""numpy restatement of the reference's EMA update and of the AdamW step its training loop runs (TEST INFRASTRUCTURE ONLY:
imported by tests/ alone; the product path is slak_amd/csrc/optim_ema.hip).

Follows:
  * ``ModelEma.update``         model_sema.py:67-91 -- every product / sum is its own float32 torch kernel there, hence its own
                                rounding here; the scalars reach those kernels as float32(decay) and float32(1. - decay), the
                                latter evaluated in double by Python.
  * ``torch.optim.AdamW.step``  the optimizer optim_factory.py:149-150 constructs.  Third-party: PyTorch (2.10 in this image),
                                torch/optim/adam.py ``_single_tensor_adam`` with decoupled weight decay; restated from its
                                published algorithm.  Then ``Masking.apply_mask`` (sparse_core.py:316-333): ``w = w * mask``.

Pinned by tests/test_oracle.py against tests/golden/ema_{masked,dense}.npz (the reference's ModelEma, imported unmodified, run by
tests/golden/make_golden.py) -- bit-exact -- and against tests/golden/mask_onlyL_adamw.npz (torch.optim.AdamW + the reference's
Masking on CPU) for the optimizer step.
"""
import numpy as np

f32 = np.float32


def ema_update_entry(ema, model, decay, mask=None):
    """One state-dict entry.  ``ema`` / ``model``: float32 or int64 arrays; ``mask``: float32 0/1 array or None."""
    d, omd = f32(decay), f32(1.0 - float(decay))
    if ema.dtype == np.int64:
        # int64 tensor * Python float -> float32 tensor; copy_ back into the int64 buffer truncates (model_sema.py:81 / :91)
        r = ema.astype(f32) * d + omd * model.astype(f32)
        return np.trunc(r).astype(np.int64)
    e, w = ema.astype(f32, copy=False), model.astype(f32, copy=False)
    if mask is None:
        return (e * d + omd * w).astype(f32)                                   # model_sema.py:81 / :91
    mb = mask.astype(np.uint8)
    diff = (((e != 0).astype(np.uint8)) ^ mb) & mb                             # model_sema.py:83-84
    out = (e * d + w * omd).astype(f32) * mask.astype(f32)                     # :86-87
    return (out + (diff.astype(f32) * d) * w).astype(f32)                      # :88


def ema_update(ema_state, model_state, decay, masks=None):
    """Dicts key -> array; returns the new EMA dict (model_sema.py:72-91: entries whose key is in ``masks`` take the sparse rule)."""
    masks = masks or {}
    return {k: ema_update_entry(v, model_state[k], decay, masks.get(k)) for k, v in ema_state.items()}


def adamw_step(p, g, m, v, step, lr, beta1, beta2, eps, weight_decay, mask=None):
    """One parameter, float32 arrays; ``step`` is the count AFTER this step's increment.  Returns (p, m, v)."""
    p, g, m, v = (a.astype(f32, copy=True) for a in (p, g, m, v))
    if weight_decay != 0:
        p = p * f32(1.0 - lr * weight_decay)                                  # param.mul_(1 - lr * weight_decay)
    m = m + f32(1.0 - beta1) * (g - m)                                        # exp_avg.lerp_(grad, 1 - beta1)
    v = v * f32(beta2)
    v = v + (f32(1.0 - beta2) * g) * g                                        # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    denom = np.sqrt(v) / f32(bc2 ** 0.5) + f32(eps)                           # (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p + f32(-step_size) * (m / denom)                                     # param.addcdiv_(exp_avg, denom, value=-step_size)
    if mask is not None:
        p = p * mask.astype(f32)                                              # sparse_core.py:326
    return p.astype(f32), m.astype(f32), v.astype(f32)
