"""Sparsity-aware exponential moving average of the model, one HIP launch per update.

Mirrors ``model_sema.ModelEma`` of the reference (model_sema.py:14-91): same constructor, same ``.ema`` / ``.decay`` attributes,
same ``update(model, mask)`` called by the training loop after every optimizer step (engine.py:75-76, :87-88).  The reference walks
the ~600 state-dict entries in Python and spends 4 (dense) to 11 (masked entry) elementwise kernels on each; here the entries are
described once to ``slak_ema_plan_create`` and every update is ``slak_ema_update``: one streaming pass over all of them, with each
product and sum rounded separately so the numbers are bit-identical to the reference's.

    dense entry  : ema = ema * decay + (1 - decay) * model
    masked entry : ema = (ema * decay + model * (1 - decay)) * mask + (diff * decay) * model,
                   diff = ((ema != 0) ^ mask) & mask           -- weights that were (re)grown since the last update start from the
                                                                  model's value instead of decaying up from zero

No CPU fallback: an EMA kept on another device than the model (``--model_ema_force_cpu``) raises.
"""
import ctypes
import logging
from collections import OrderedDict
from copy import deepcopy

import torch

from . import _lib
from .ops import _on, _stream

_logger = logging.getLogger(__name__)
_SLAK_I64 = 3


class ModelEma:
    def __init__(self, model, decay=0.9999, device='', resume=''):
        self.ema = deepcopy(model)
        self.ema.eval()
        self.decay = decay
        self.device = device
        if device:
            self.ema.to(device=device)
        self.ema_has_module = hasattr(self.ema, 'module')
        if resume:
            self._load_checkpoint(resume)
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self._plan = None
        self._plan_key = None
        self._tensors = None

    def _load_checkpoint(self, checkpoint_path):
        """model_sema.py:46-65"""
        checkpoint = torch.load(checkpoint_path, map_location='cpu')
        assert isinstance(checkpoint, dict)
        if 'state_dict_ema' in checkpoint:
            new_state_dict = OrderedDict()
            for k, v in checkpoint['state_dict_ema'].items():
                if self.ema_has_module:
                    name = 'module.' + k if not k.startswith('module') else k
                else:
                    name = k
                new_state_dict[name] = v
            self.ema.load_state_dict(new_state_dict)
            _logger.info("Loaded state_dict_ema")
        else:
            _logger.warning("Failed to find state_dict_ema, starting from loaded model weights")

    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _entries(module):
        """(key, tensor) for everything state_dict() would hold, without building the detached views each step."""
        out = []
        for prefix, m in module.named_modules():
            for name, t in list(m._parameters.items()) + [(n, b) for n, b in m._buffers.items() if n not in m._non_persistent_buffers_set]:
                if t is not None:
                    out.append(((prefix + '.' if prefix else '') + name, t))
        return out

    def _ensure_plan(self, model, mask):
        if self._tensors is None or self._tensors[0] is not model:
            needs_module = hasattr(model, 'module') and not self.ema_has_module
            msd = dict(self._entries(model))
            rows = []
            for k, ema_v in self._entries(self.ema):
                mk = 'module.' + k if needs_module else k
                rows.append((mk, ema_v, msd[mk]))                              # KeyError like the reference's msd[k]
            self._tensors = (model, rows)
            self._plan_key = None
        rows = self._tensors[1]
        masks = mask.masks if mask else {}
        key = tuple((e.data_ptr(), w.data_ptr(), masks[k].data_ptr() if k in masks else 0) for k, e, w in rows)
        if key == self._plan_key:
            return
        self._destroy_plan()
        live = [(k, e, w) for k, e, w in rows if e.numel() > 0]
        segs = (_lib.EmaSegment * max(len(live), 1))()
        for i, (k, e, w) in enumerate(live):
            if not e.is_cuda or e.device != w.device:
                raise _lib.SlakHipError("slak_amd ModelEma needs the EMA and the model on the same HIP device (entry %s: %s vs %s); "
                                        "there is no CPU fallback" % (k, e.device, w.device))
            if e.dtype != w.dtype or e.shape != w.shape or not e.is_contiguous() or not w.is_contiguous():
                raise _lib.SlakHipError("EMA entry %s does not match the model's (dtype/shape/contiguity)" % k)
            if e.dtype == torch.float32:
                dt = _lib.SLAK_F32
            elif e.dtype == torch.int64:
                dt = _SLAK_I64
            else:
                raise _lib.SlakHipError("EMA entry %s has dtype %s; float32 and int64 entries are supported" % (k, e.dtype))
            m = masks.get(k)
            if m is not None and (m.dtype != torch.float32 or m.device != e.device or m.shape != e.shape or not m.is_contiguous() or dt != _lib.SLAK_F32):
                raise _lib.SlakHipError("mask of %s must be a contiguous float32 tensor of the entry's shape on its device" % k)
            segs[i].ema, segs[i].model = e.data_ptr(), w.data_ptr()
            segs[i].mask = m.data_ptr() if m is not None else None
            segs[i].numel, segs[i].dtype = e.numel(), dt
        if live:
            h = ctypes.c_void_p()
            with _on(live[0][1].device):
                _lib.check(_lib.lib().slak_ema_plan_create(segs, len(live), ctypes.byref(h)), "slak_ema_plan_create")
            self._plan = (h, live[0][1].device)
        self._plan_key = key

    def _destroy_plan(self):
        if self._plan is not None:
            _lib.lib().slak_ema_plan_destroy(self._plan[0])
            self._plan = None

    def __del__(self):
        try:
            self._destroy_plan()
        except Exception:
            pass

    def update(self, model, mask):
        """model_sema.py:67-91.  ``mask``: the slak_amd.sparse_core.Masking object of the run, or None / False for a dense EMA."""
        with torch.no_grad():
            self._ensure_plan(model, mask)
            if self._plan is None:
                return
            h, dev = self._plan
            with _on(dev):
                _lib.check(_lib.lib().slak_ema_update(h, float(self.decay), _stream(dev)), "slak_ema_update")
