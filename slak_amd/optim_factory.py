"""Optimizer construction for the SLaK training loop, with the AdamW update, Masking's ``w *= mask`` and the bf16 weight copies
of the pointwise GEMMs as ONE HIP launch per step.

Mirrors the surface of the reference's ``optim_factory.py``: ``create_optimizer`` (:115-199), ``get_parameter_groups`` (:73-112),
``LayerDecayValueAssigner`` / ``get_num_layer_for_convnext`` (:32-70).  ``--opt adamw`` (the default of main.py and of every README
recipe) returns :class:`MaskedAdamW`; the plain torch.optim choices are passed through; the timm / apex optimizers the reference can
also name are not part of this path and raise.

There is no CPU fallback: MaskedAdamW.step() needs its parameters on a HIP device and libslak_hip.so built.
"""
import ctypes
import json

import torch
from torch import optim

from . import _lib
from .ops import _on, _stream


# ------------------------------------------------------------------------------------------------------------------ layer decay
def get_num_layer_for_convnext(var_name):
    """Layer id of a parameter for layer-wise lr decay: 12 groups over the [3, 3, 27, 3] ConvNeXt layout, three consecutive
    stage-3 blocks per group, downsample layers attached to the neighbouring group (optim_factory.py:32-59)."""
    num_max_layer = 12
    parts = var_name.split('.')
    if var_name.startswith("downsample_layers"):
        return {0: 0, 1: 2, 2: 3, 3: 12}[int(parts[1])]
    if var_name.startswith("stages"):
        stage_id, block_id = int(parts[1]), int(parts[2])
        if stage_id in (0, 1):
            return stage_id + 1
        if stage_id == 2:
            return 3 + block_id // 3
        return 12
    return num_max_layer + 1


class LayerDecayValueAssigner(object):
    """optim_factory.py:61-70"""

    def __init__(self, values):
        self.values = values

    def get_scale(self, layer_id):
        return self.values[layer_id]

    def get_layer_id(self, var_name):
        return get_num_layer_for_convnext(var_name)


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None):
    """One group per {decay, no_decay} (x layer id under layer decay).  1-D parameters, ``*.bias`` and the skip list get no
    weight decay; every group carries ``lr_scale`` (engine.py:41-46 multiplies the scheduled lr by it).  optim_factory.py:73-112."""
    names, groups = {}, {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        no_decay = param.dim() == 1 or name.endswith(".bias") or name in skip_list
        group_name = "no_decay" if no_decay else "decay"
        layer_id = None
        if get_num_layer is not None:
            layer_id = get_num_layer(name)
            group_name = "layer_%d_%s" % (layer_id, group_name)
        if group_name not in groups:
            scale = get_layer_scale(layer_id) if get_layer_scale is not None else 1.
            wd = 0. if no_decay else weight_decay
            names[group_name] = {"weight_decay": wd, "params": [], "lr_scale": scale}
            groups[group_name] = {"weight_decay": wd, "params": [], "lr_scale": scale}
        groups[group_name]["params"].append(param)
        names[group_name]["params"].append(name)
    print("Param groups = %s" % json.dumps(names, indent=2))
    return list(groups.values())


# ------------------------------------------------------------------------------------------------------------------ the optimizer
class MaskedAdamW(optim.Optimizer):
    """torch.optim.AdamW semantics (decoupled weight decay; same state keys ``step`` / ``exp_avg`` / ``exp_avg_sq``, so optimizer
    checkpoints interchange) with the whole step as one kernel over all parameters: ``slak_adamw_step`` (include/slak_hip.h).

    * ``set_masks({param: mask})`` (called by slak_amd.sparse_core.Masking) folds ``Masking.apply_mask``'s ``w *= mask``
      (sparse_core.py:316-333) into the same pass; ``applies_masks`` tells Masking.step() to skip its own apply launch.
    * when ``slak_amd.block_ops.cache_lowp_weights`` is on, the bf16 copies the pointwise GEMMs read are written by the same pass.
    * step counts live in one device buffer (``state[p]['step']`` are 0-dim views of it, like torch's fused/capturable AdamW);
      the bias corrections are evaluated on device in double, as Python does.
    """

    applies_masks = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError("invalid AdamW hyper-parameters")
        # the extra keys are torch.optim.AdamW's own group entries at their defaults, so a saved state dict loads into either class
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=None,
                                      capturable=False, differentiable=False, fused=None, decoupled_weight_decay=True))
        self._masks = {}
        self._plans = None          # list of (plan handle, [params], first group index, number of groups)
        self._key = None
        self._steps = None
        self._state_dirty = True
        self._grad_key = None
        self._grad_tabs = None
        self._active = None         # per-parameter 0/1 step increments when some gradients are missing (None: all present)

    # -- Masking hook ---------------------------------------------------------------------------------------------
    def set_masks(self, masks):
        """masks: {parameter: fp32 0/1 tensor of the same shape} (or None to detach)."""
        self._masks = {id(p): m for p, m in (masks or {}).items()}
        self._key = None

    # -- state ----------------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._state_dirty = True
        self._key = None
        params = self._all_params()
        if params and all(p.is_cuda for _, p in params):
            self._ensure_state(params)        # take the step counts NOW: torch hands the caller's own 'step' tensors through uncopied

    def _all_params(self):
        out = []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.requires_grad:
                    out.append((gi, p))
        return out

    def _ensure_state(self, params):
        dev = params[0][1].device
        for _, p in params:
            if not p.is_cuda or p.device != dev:
                raise _lib.SlakHipError("MaskedAdamW needs every parameter on one HIP device (got %s); there is no CPU fallback" % (p.device,))
            if p.dtype != torch.float32 or not p.data.is_contiguous():
                raise _lib.SlakHipError("MaskedAdamW parameters must be contiguous float32")
        if self._steps is None or self._steps.numel() != len(params) or self._steps.device != dev:
            self._steps = torch.zeros(len(params), dtype=torch.float32, device=dev)
            self._state_dirty = True
        if not self._state_dirty:
            return
        for i, (_, p) in enumerate(params):
            st = self.state[p]
            slot = self._steps[i]
            old = st.get("step")
            if old is not None and not (torch.is_tensor(old) and old.data_ptr() == slot.data_ptr()):
                slot.copy_(torch.as_tensor(old, dtype=torch.float32))           # loaded from a checkpoint (torch keeps it on the CPU)
            st["step"] = slot
            for k in ("exp_avg", "exp_avg_sq"):
                t = st.get(k)
                if t is None or t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.shape != p.shape:
                    st[k] = torch.zeros_like(p, memory_format=torch.contiguous_format) if t is None else \
                        t.to(device=dev, dtype=torch.float32).contiguous().view_as(p)
        self._state_dirty = False
        self._key = None

    def _ensure_plans(self, params):
        from . import block_ops
        lowp = block_ops._lowp_cache if block_ops.cache_lowp_weights else {}
        gen = len(lowp)
        key = (tuple(p.data_ptr() for _, p in params), gen, len(self.param_groups))
        if self._plans is not None and key == self._key:
            return
        self._destroy_plans()
        L = _lib.lib()
        plans = []
        G = _lib.ADAMW_MAX_GROUPS
        for g0 in range(0, len(self.param_groups), G):
            sel = [(i, gi, p) for i, (gi, p) in enumerate(params) if g0 <= gi < g0 + G]
            if not sel:
                continue
            segs = (_lib.AdamwSegment * len(sel))()
            cached = []
            for j, (i, gi, p) in enumerate(sel):
                st = self.state[p]
                m = self._masks.get(id(p))
                if m is not None and (m.device != p.device or m.dtype != torch.float32 or m.numel() != p.numel() or not m.is_contiguous()):
                    raise _lib.SlakHipError("mask of a parameter must be a contiguous float32 tensor of its shape on its device")
                e = lowp.get(id(p))
                if e is not None and (e[0]() is not p or e[2].device != p.device or not e[2].is_contiguous()):
                    e = None
                segs[j].param = p.data_ptr()
                segs[j].exp_avg = st["exp_avg"].data_ptr()
                segs[j].exp_avg_sq = st["exp_avg_sq"].data_ptr()
                segs[j].mask = m.data_ptr() if m is not None else None
                segs[j].param_bf16 = e[2].data_ptr() if e is not None else None
                segs[j].step = st["step"].data_ptr()
                segs[j].numel = p.numel()
                segs[j].group = gi - g0
                if e is not None:
                    cached.append((p, e))
            h = ctypes.c_void_p()
            with _on(params[0][1].device):
                _lib.check(L.slak_adamw_plan_create(segs, len(sel), ctypes.byref(h)), "slak_adamw_plan_create")
            plans.append((h, [p for _, _, p in sel], g0, min(G, len(self.param_groups) - g0), cached))
        self._plans, self._key = plans, key
        self._grad_key = None

    def _destroy_plans(self):
        if self._plans:
            L = _lib.lib()
            for h, *_ in self._plans:
                L.slak_adamw_plan_destroy(h)
        self._plans = None

    def __del__(self):
        try:
            self._destroy_plans()
        except Exception:
            pass

    # -- the step -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        params = self._all_params()
        if not params:
            return loss
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("decoupled_weight_decay") is False:
                raise NotImplementedError("MaskedAdamW implements AdamW (decoupled weight decay) without amsgrad / maximize")
        self._ensure_state(params)
        self._ensure_plans(params)
        dev = params[0][1].device
        grads = [p.grad for _, p in params]
        gkey = tuple(0 if g is None else g.data_ptr() for g in grads)
        from . import block_ops
        if block_ops.grad_slots_enabled:                              # DDP bucket views: the next backward writes its gradients straight into them
            block_ops.adopt_grad_slots([p for _, p in params], key=gkey)
        if gkey != self._grad_key:
            for g, (_, p) in zip(grads, params):
                if g is not None and (g.is_sparse or g.dtype != torch.float32 or not g.is_contiguous() or g.device != dev or g.shape != p.shape):
                    raise _lib.SlakHipError("MaskedAdamW gradients must be dense contiguous float32 tensors on the parameters' device")
            tabs, at = [], 0
            for h, ps, *_ in self._plans:
                host = torch.tensor(gkey[at:at + len(ps)], dtype=torch.int64).pin_memory()
                tabs.append(host.to(dev, non_blocking=True))
                at += len(ps)
            self._grad_tabs, self._grad_key = tabs, gkey
            self._active = None if all(gkey) else torch.tensor([1.0 if k else 0.0 for k in gkey], device=dev)
        if self._active is None:
            self._steps.add_(1.0)
        else:
            self._steps.add_(self._active)
        L = _lib.lib()
        stream = _stream(dev)
        with _on(dev):
            for (h, ps, g0, ng, cached), tab in zip(self._plans, self._grad_tabs):
                hyp = (_lib.AdamwGroup * ng)()
                for k in range(ng):
                    g = self.param_groups[g0 + k]
                    hyp[k].lr, hyp[k].beta1, hyp[k].beta2 = float(g["lr"]), float(g["betas"][0]), float(g["betas"][1])
                    hyp[k].eps, hyp[k].weight_decay = float(g["eps"]), float(g["weight_decay"])
                _lib.check(L.slak_adamw_step(h, tab.data_ptr(), hyp, ng, stream), "slak_adamw_step")
        # the kernel wrote through raw pointers: bump the autograd version counters, and mark the bf16 copies it refreshed as current
        from .sparse_core import Masking
        Masking._bump_versions([p for (_, p), k in zip(params, gkey) if k])
        for _, _, _, _, cached in self._plans:
            for p, e in cached:
                if p.grad is not None:
                    e[1] = p._version
        return loss


# ------------------------------------------------------------------------------------------------------------------ factory
_TIMM_OR_APEX = ("nadam", "radam", "adamp", "sgdp", "adafactor", "adahessian", "rmsproptf", "nvnovograd", "fusedsgd", "fusedmomentum",
                 "fusedadam", "fusedadamw", "fusedlamb", "fusednovograd")


def create_optimizer(args, model, get_num_layer=None, get_layer_scale=None, filter_bias_and_bn=True, skip_list=None):
    """optim_factory.py:115-199.  ``args``: opt, lr, weight_decay, momentum, and optionally opt_eps / opt_betas."""
    opt_lower = args.opt.lower()
    weight_decay = args.weight_decay
    if filter_bias_and_bn:
        skip = {}
        if skip_list is not None:
            skip = skip_list
        elif hasattr(model, 'no_weight_decay'):
            skip = model.no_weight_decay()
        parameters = get_parameter_groups(model, weight_decay, skip, get_num_layer, get_layer_scale)
        weight_decay = 0.
    else:
        parameters = model.parameters()
    opt_args = dict(lr=args.lr, weight_decay=weight_decay)
    if getattr(args, 'opt_eps', None) is not None:
        opt_args['eps'] = args.opt_eps
    if getattr(args, 'opt_betas', None) is not None:
        opt_args['betas'] = args.opt_betas
    opt_split = opt_lower.split('_')
    opt_lower = opt_split[-1]
    if len(opt_split) > 1:
        raise NotImplementedError("optimizer wrapper %r (timm Lookahead) is outside the MI355X hot path" % opt_split[0])
    if opt_lower in ('sgd', 'nesterov'):
        opt_args.pop('eps', None)
        return optim.SGD(parameters, momentum=args.momentum, nesterov=True, **opt_args)
    if opt_lower == 'momentum':
        opt_args.pop('eps', None)
        return optim.SGD(parameters, momentum=args.momentum, nesterov=False, **opt_args)
    if opt_lower == 'adam':
        return optim.Adam(parameters, **opt_args)
    if opt_lower == 'adamw':
        return MaskedAdamW(parameters, **opt_args)
    if opt_lower == 'adadelta':
        return optim.Adadelta(parameters, **opt_args)
    if opt_lower == 'rmsprop':
        return optim.RMSprop(parameters, alpha=0.9, momentum=args.momentum, **opt_args)
    if opt_lower in _TIMM_OR_APEX:
        raise NotImplementedError("--opt %s is a timm/apex optimizer of the reference (optim_factory.py:151-190); not part of this path" % opt_lower)
    raise ValueError("Invalid optimizer %r" % args.opt)
