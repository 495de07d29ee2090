"""Drop-in for the reference's ``sparse_core`` module (sparse_core.py:49-407): ``CosineDecay`` and
``Masking`` with the same constructor, attributes and methods, but with the per-step and per-update
work done by batched HIP kernels (slak_amd/csrc/mask_kernels.hip) over ALL masked tensors at once:

    reference (per tensor, Python)                         here (all tensors, on device)
    -----------------------------------------------------  -----------------------------------------
    apply_mask: w.data = w.data*mask           :316-333     slak_mask_apply          (1 launch)
    truncate_weights: mask.sum().item(),       :335-357     slak_mask_prune_and_grow (no host sync;
      torch.sort(|w|), torch.sort(|grad|), ...                 radix select, counts stay on device)
    synchronism_masks: broadcast every mask    :404-407     ONE broadcast after init; afterwards masks
      from rank 0, every step                                 are a deterministic function of replicated
                                                              weights + all-reduced grads (checksum
                                                              all-reduce available as a debug check)

Scope (BASELINE.json north star): growth='gradient', prune='magnitude' -- all on device.  growth='random'
(funcs.random_growth, funcs.py:170-175; the default of main.py:211 and what the README recipes pass) is supported too: the
prune half runs on device (slak_mask_prune), the random numbers are the HOST generator's, as in the reference
(``torch.rand(shape).cuda() < p``), so a seed reproduces the reference's masks; because ranks draw different numbers the masks
are re-broadcast from rank 0 after every such growth (the reference broadcasts them every step).  The remaining growth /
prune / redistribution modes of funcs.py raise NotImplementedError rather than silently falling back.  Mask initialisation (uniform / resume / snip / ERK, sparse_core.py:141-261) is one-time
host logic and stays in Python, using the same torch RNG calls so that seeds reproduce the reference.

Masks stay fp32 tensors in ``self.masks[name]`` (``model_sema.py:83-89`` reads them), keyed by
``named_parameters()`` names of whatever was passed to ``add_module`` (the DDP wrapper in main.py:425,
so keys carry ``module.``).  Ties at the prune/grow cut: lowest flat index first (torch.sort(stable=True));
the reference's unstable torch.sort is arbitrary there.
"""
from __future__ import print_function

import ctypes
import math

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim

from . import _lib
from .ops import _on, _stream

__all__ = ["CosineDecay", "Masking", "SNIP"]


def SNIP(net, keep_ratio, train_dataloader, device, masks, args):
    """Layer-wise sparsities from one batch of |w * dL/dw| saliencies (sparse_core.py:11-47)."""
    import copy
    if args.distributed:
        train_dataloader.sampler.set_epoch(0)
    images, labels = next(iter(train_dataloader))
    images = images.to(device, non_blocking=True)
    labels = labels.to(device, non_blocking=True)
    probe = copy.deepcopy(net)            # never touch the training copy
    probe.zero_grad()
    F.cross_entropy(probe(images), labels).backward()
    saliency = [torch.abs(p * p.grad) for n, p in probe.named_parameters() if n in masks]
    flat = torch.cat([s.flatten() for s in saliency])
    keep = int(len(flat) * keep_ratio)
    cut = torch.topk(flat, keep, sorted=True)[0][-1]
    out = []
    for s in saliency:
        kept = (s > cut).float()
        out.append(float((kept == 0).sum().item() / kept.numel()))
    probe.zero_grad()
    return out


class CosineDecay(object):
    """Prune-rate schedule: torch's CosineAnnealingLR on a dummy SGD, exactly as the reference builds it
    (sparse_core.py:49-64), kept on the host so ``ceil(rate*nnz)`` sees the bit-identical fp64 rate."""

    def __init__(self, prune_rate, T_max, eta_min=0.005, last_epoch=-1, init_step=0):
        self.sgd = optim.SGD(torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(1))]), lr=prune_rate)
        self.cosine_stepper = torch.optim.lr_scheduler.CosineAnnealingLR(self.sgd, T_max, eta_min, last_epoch)
        for _ in range(init_step):
            self.cosine_stepper.step()

    def step(self):
        self.cosine_stepper.step()

    def get_dr(self, prune_rate):
        return self.sgd.param_groups[0]['lr']


class Masking(object):
    """Same surface as the reference class (sparse_core.py:67-407).

    Basic usage (unchanged from the reference):
        decay = CosineDecay(args.prune_rate, len(train_loader)*args.epochs)
        mask = Masking(optimizer, train_loader, prune_rate_decay=decay, prune_rate=args.prune_rate,
                       prune_mode='magnitude', growth_mode='gradient', redistribution_mode='none', args=args)
        mask.add_module(model)
        ...
        mask.step()        # instead of optimizer.step()
    """

    def __init__(self, optimizer, train_loader, prune_rate_decay, prune_rate=0.5, prune_mode='magnitude',
                 growth_mode='random', redistribution_mode='momentum', verbose=False, fp16=False, args=False):
        self.args = args
        self.device = torch.device(args.device)
        self.growth_mode = growth_mode
        self.prune_mode = prune_mode
        self.redistribution_mode = redistribution_mode
        self.prune_rate_decay = prune_rate_decay
        self.verbose = verbose
        self.train_loader = train_loader
        self.growth_func = growth_mode
        self.prune_func = prune_mode
        self.redistribution_func = redistribution_mode
        self.global_growth = False
        self.global_prune = False

        self.masks = {}
        self.modules = []
        self.names = []
        self.optimizer = optimizer
        self.baseline_nonzero = None

        self.name2zeros = {}
        self.name2nonzeros = {}
        self.name2removed = {}
        self.prune_rate = prune_rate
        self.steps = 0
        self.half = fp16
        self.name_to_32bit = {}

        if self.args.fix:
            self.args.update_frequency = None

        # device plan state
        self._plan = None
        self._plan_names = []
        self._plan_key = None
        self._synced_once = False
        # rank agreement: masks are computed independently on every rank after the one broadcast, so a rank-local perturbation
        # (un-reduced gradients under no_sync(), a rank-local weight edit, a nondeterministic kernel) would otherwise diverge
        # silently; the reference re-broadcasts every step and so heals itself (sparse_core.py:404-407).  One 24-byte all-reduce
        # per prune-and-grow round / load_state_dict (every update_frequency steps) checks it; on a mismatch rank 0's masks are
        # re-broadcast (the reference's rule) and counted in `rank_resyncs`, or an error is raised with args.debug_mask_sync.
        self.check_ranks = bool(getattr(args, "check_mask_sync", True))
        self.debug_check_ranks = bool(getattr(args, "debug_mask_sync", False))
        self.rank_resyncs = 0

        if self.half:
            raise NotImplementedError("fp16 master-copy masking (apex FP16_Optimizer, sparse_core.py:135-139, :329-333) "
                                      "is outside the MI355X hot path; use bf16 autocast with fp32 parameters")

    # ------------------------------------------------------------------ construction / init
    def add_module(self, module):
        self.modules.append(module)
        self.module = module
        self._params_cache = None
        from . import block_ops
        block_ops.enable_grad_slots_for(module)                      # main.py:425 passes the DDP wrapper: bucket-view gradients are written in place
        for name, tensor in module.named_parameters():
            if tensor.dim() in (2, 4):
                if self.args.only_L and 'large_kernel.LoRA' not in name:
                    continue
                self.names.append(name)
                self.masks[name] = torch.zeros_like(tensor, dtype=torch.float32, requires_grad=False).to(self.device)
        self.init(mode=self.args.sparse_init, density=1 - self.args.sparsity)

    def init_optimizer(self):
        if 'fp32_from_fp16' in self.optimizer.state_dict():
            raise NotImplementedError("fp16 master-copy optimizers are not supported")

    def init_growth_prune_and_redist(self):
        if self.growth_mode not in ('gradient', 'random'):
            raise NotImplementedError("growth mode %r: 'gradient' (funcs.py:196-205, all on device) and 'random' (funcs.py:170-175, the "
                                      "default of main.py:211; its random numbers come from the HOST generator) are supported" % (self.growth_mode,))
        if self.prune_mode != 'magnitude':
            raise NotImplementedError("prune mode %r: only 'magnitude' (funcs.py:107-114) is on the MI355X hot path" % (self.prune_mode,))
        # the reference resolves a redistribution function but never calls it (sparse_core.py:288-297)

    def _rand_mask(self, shape, density):
        # same RNG stream as the reference: CPU torch.rand, then moved (sparse_core.py:155, :180, :241)
        return (torch.rand(shape) < density).float().data.to(self.device)

    def init(self, mode='snip', density=0.05, erk_power_scale=1.0):
        self.init_growth_prune_and_redist()
        self.init_optimizer()
        self.density = density

        if mode == 'uniform':
            print('initialized with uniform')
            self.baseline_nonzero = 0
            for module in self.modules:
                for name, weight in module.named_parameters():
                    if name not in self.masks:
                        continue
                    self.masks[name][:] = self._rand_mask(weight.shape, density)
                    self.baseline_nonzero += weight.numel() * density
        elif mode == 'resume':
            print('initialized with resume')
            self.baseline_nonzero = 0
            for module in self.modules:
                for name, weight in module.named_parameters():
                    if name not in self.masks:
                        continue
                    print((weight != 0.0).sum().item())
                    self.masks[name][:] = (weight != 0.0).float().data.to(self.device)
                    self.baseline_nonzero += weight.numel() * density
        elif mode == 'snip':
            print('initialize by snip')
            self.baseline_nonzero = 0
            sparsities = SNIP(self.module, density, self.train_loader, self.device, self.masks, self.args)
            for sp, name in zip(sparsities, self.masks):
                self.masks[name][:] = self._rand_mask(self.masks[name].shape, 1 - sp)
        elif mode == 'ERK':
            print('initialize by fixed_ERK')
            self._init_erk(density, erk_power_scale)
        else:
            raise ValueError("unknown sparse_init %r" % (mode,))

        total_size = sparse_size = 0
        dense_layers = []
        for name, m in self.masks.items():
            n_all = m.numel()
            n_on = (m != 0).sum().int().item()
            total_size += n_all
            sparse_size += n_on
            layer_density = n_on / n_all
            if layer_density >= 0.99:
                dense_layers.append(name)
            print(f'Density of layer {name} with tensor {m.size()} is {layer_density}')
        print('Final sparsity level of {0}: {1}'.format(1 - self.density, 1 - sparse_size / total_size))
        for name in dense_layers:                       # (almost) dense layers are not masked at all
            self.masks.pop(name)
            print(f"pop out layer {name}")
        self._plan_key = None
        self._synced_once = False
        self.apply_mask()

    def _init_erk(self, density, erk_power_scale):
        """Erdos-Renyi-Kernel densities: layer probability ~ (sum of dims / prod of dims)^scale, scaled by a
        global epsilon so the total budget is met; layers whose probability would exceed 1 become dense and
        the rest is re-solved (sparse_core.py:183-245)."""
        total_params = sum(m.numel() for m in self.masks.values())
        self.baseline_nonzero = sum(m.numel() * density for m in self.masks.values())
        dense = set()
        while True:
            divisor, rhs, raw = 0, 0, {}
            for name, m in self.masks.items():
                n_param = np.prod(m.shape)
                if name in dense:
                    rhs -= n_param * (1 - density)
                else:
                    rhs += n_param * density
                    raw[name] = (np.sum(m.shape) / np.prod(m.shape)) ** erk_power_scale
                    divisor += raw[name] * n_param
            epsilon = rhs / divisor
            top = np.max(list(raw.values()))
            if top * epsilon > 1:
                for name, r in raw.items():
                    if r == top:
                        print(f"Sparsity of var:{name} had to be set to 0.")
                        dense.add(name)
            else:
                break
        total_nonzero = 0.0
        for name, m in self.masks.items():
            d = 1.0 if name in dense else epsilon * raw[name]
            print(f"layer: {name}, shape: {m.shape}, density: {d}")
            self.masks[name][:] = self._rand_mask(m.shape, d)
            total_nonzero += d * m.numel()
        print(f"Overall sparsity {total_nonzero / total_params}")

    # ------------------------------------------------------------------ device plan
    def _masked_params(self, refresh=False):
        """[(name, parameter)] of the masked tensors in module order.  Walking named_parameters() of a DDP-wrapped SLaK costs ~1.2 ms of host time, and step()
        needs the list twice: it is kept and re-derived when the mask set changes (add_module / remove_* drop it), on a prune-and-grow round, on
        load_state_dict(), or when a kept entry is no longer the module's own: every kept tensor is checked BY IDENTITY against the live attribute of the
        module that owns it (a Parameter replaced on the same device -- `m.weight = nn.Parameter(...)` -- is a different object) and against its mask's
        device."""
        cache = getattr(self, "_params_cache", None)
        if not refresh and cache is not None and cache[0] == len(self.masks) and cache[1] == len(self.modules):
            ok = True
            for (n, t), (owner, attr) in zip(cache[2], cache[3]):
                m = self.masks.get(n)
                if m is None or owner._parameters.get(attr) is not t or t.device != m.device:
                    ok = False
                    break
            if ok:
                return cache[2]
        out, owners = [], []
        for module in self.modules:
            subs = dict(module.named_modules())
            for name, tensor in module.named_parameters():
                if name in self.masks:
                    prefix, _, attr = name.rpartition(".")
                    out.append((name, tensor))
                    owners.append((subs[prefix], attr))
        self._params_cache = (len(self.masks), len(self.modules), out, owners)
        return out

    def _ensure_plan(self):
        params = self._masked_params()
        if not params:
            return None
        for name, t in params:
            if not t.is_cuda:
                raise _lib.SlakHipError("slak_amd Masking needs parameters on a HIP device (got %s for %s); there is no CPU fallback" % (t.device, name))
            if t.dtype != torch.float32 or not t.data.is_contiguous():
                raise _lib.SlakHipError("masked parameter %s must be contiguous float32" % name)
        key = tuple((n, t.data.data_ptr(), self.masks[n].data_ptr(), t.numel()) for n, t in params)
        if key != self._plan_key:
            L = _lib.lib()
            if self._plan is not None:
                L.slak_mask_plan_destroy(self._plan)
                self._plan = None
            segs = (_lib.MaskSegment * len(params))()
            for i, (n, t) in enumerate(params):
                segs[i].weight = t.data.data_ptr()
                segs[i].mask = self.masks[n].data_ptr()
                segs[i].grad = None
                segs[i].momentum = None
                segs[i].numel = t.numel()
            plan = ctypes.c_void_p()
            with _on(self.device):
                _lib.check(L.slak_mask_plan_create(segs, len(params), ctypes.byref(plan)), "slak_mask_plan_create")
            self._plan, self._plan_key = plan, key
            self._plan_names = [n for n, _ in params]
            self._momentum_key = None
        return params

    def _stream(self):
        return _stream(self.device)

    def _bind_momentum(self, params):
        ptrs = []
        for _, t in params:
            st = self.optimizer.state.get(t, {}) if self.optimizer is not None else {}
            mb = st.get('momentum_buffer') if isinstance(st, dict) else None
            ptrs.append(mb.data_ptr() if (mb is not None and mb.is_cuda and mb.dtype == torch.float32 and mb.is_contiguous()) else 0)
        key = tuple(ptrs)
        if key != self._momentum_key:
            arr = (ctypes.c_void_p * len(ptrs))(*[p or None for p in ptrs])
            _lib.check(_lib.lib().slak_mask_plan_set_momentum(self._plan, arr, self._stream()), "slak_mask_plan_set_momentum")
            self._momentum_key = key

    # ------------------------------------------------------------------ the per-step surface
    def _bind_optimizer_masks(self):
        """An optimizer that can apply masks inside its own update kernel (slak_amd.optim_factory.MaskedAdamW) gets the mask of
        every masked parameter; returns True when it took them, i.e. when optimizer.step() already leaves w == w * mask."""
        if not getattr(self.optimizer, "applies_masks", False):
            return False
        params = self._masked_params()
        key = tuple((id(t), self.masks[n].data_ptr()) for n, t in params)
        if key != getattr(self, "_opt_masks_key", None):
            self.optimizer.set_masks({t: self.masks[n] for n, t in params})
            self._opt_masks_key = key
        return True

    def step(self):
        if self._bind_optimizer_masks():
            if self.args.distributed:
                self.synchronism_masks()
            self.optimizer.step()                       # AdamW update and w *= mask in one launch (sparse_core.py:300-303)
        else:
            self.optimizer.step()
            self.apply_mask()
        self.prune_rate_decay.step()
        self.prune_rate = self.prune_rate_decay.get_dr(self.prune_rate)
        self.steps += 1
        if self.args.update_frequency is not None:
            if self.steps % self.args.update_frequency == 0:
                print('*********************************Dynamic Sparsity********************************')
                self.truncate_weights()
                self.print_nonzero_counts()

    def apply_mask(self):
        if self.args.distributed:
            self.synchronism_masks()
        params = self._ensure_plan()
        if params is None:
            return
        with _on(self.device):
            self._bind_momentum(params)
            _lib.check(_lib.lib().slak_mask_apply(self._plan, self._stream()), "slak_mask_apply")
        self._bump_versions([t for _, t in params])

    @staticmethod
    def _bump_versions(params):
        """The kernels write the weights through raw pointers; tell autograd (and anything that caches by tensor version, e.g. the
        bf16 weight copies of slak_amd.block_ops) that they changed."""
        inc = getattr(torch._C, "_increment_version", None)
        if inc is None or not params:
            return
        try:
            inc(list(params))                  # torch >= 2.4: an iterable of tensors (a bare tensor would be ITERATED: unbind per row)
        except TypeError:
            for p in params:                   # older torch: one tensor per call
                inc(p)

    def _truncate_weights_random_growth(self, params):
        """prune on device, then funcs.random_growth (funcs.py:170-175) with the reference's own random numbers:
        ``torch.rand(shape)`` from the HOST generator, one draw per masked tensor in the reference's loop order, compared against
        p = total_regrowth / (zeros of the pruned mask) on the device.  The counts come from the prune kernels' statistics (one
        host synchronisation per prune-and-grow round; the reference has three per tensor).  Ranks draw different numbers
        (main.py:232 seeds with seed + rank), which is what the reference's per-step mask broadcast papers over: the masks are
        re-synchronised from rank 0 right after this growth."""
        L = _lib.lib()
        with _on(self.device):
            self._bind_momentum(params)
            _lib.check(L.slak_mask_prune(self._plan, float(self.prune_rate), self._stream()), "slak_mask_prune")
            stats = (ctypes.c_double * (4 * len(params)))()
            _lib.check(L.slak_mask_read_stats(self._plan, stats, self._stream()), "slak_mask_read_stats")
        self._nonzeros_after = {}
        for i, (name, t) in enumerate(params):
            self.name2nonzeros[name] = stats[4 * i + 0]
            self.name2zeros[name] = stats[4 * i + 1]
            self.name2removed[name] = stats[4 * i + 2]
        for i, (name, t) in enumerate(params):                      # growth loop, sparse_core.py:349-355
            mask = self.masks[name]
            total_regrowth = math.floor(self.name2removed[name])
            n = int(mask.numel() - (self.name2nonzeros[name] - self.name2removed[name]))     # (new_mask == 0).sum().item()
            if n == 0:
                continue
            expeced_growth_probability = total_regrowth / n
            new_weights = torch.rand(mask.shape).to(self.device) < expeced_growth_probability
            mask.copy_((mask.bool() | new_weights).float())         # in place: the device plan and the optimizer keep their pointers
        self._synced_once = False                                   # ranks drew different numbers: rank 0's masks win (sparse_core.py:404-407)
        self.apply_mask()                                           # sparse_core.py:357

    def truncate_weights(self):
        self._masked_params(refresh=True)                # a prune-and-grow round re-derives the parameter list from the modules (see _masked_params)
        params = self._ensure_plan()
        if params is None:
            return
        if self.growth_mode == 'random':
            return self._truncate_weights_random_growth(params)
        L = _lib.lib()
        grads = []
        for name, t in params:
            g = t.grad
            if g is None:
                raise RuntimeError("truncate_weights needs .grad of %s (gradient growth, funcs.py:196-205)" % name)
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise _lib.SlakHipError("gradient of %s must be contiguous float32" % name)
            grads.append(g)
        with _on(self.device):
            self._bind_momentum(params)
            arr = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
            _lib.check(L.slak_mask_plan_set_grads(self._plan, arr, self._stream()), "slak_mask_plan_set_grads")
            _lib.check(L.slak_mask_prune_and_grow(self._plan, float(self.prune_rate), self._stream()), "slak_mask_prune_and_grow")
            stats = (ctypes.c_double * (4 * len(params)))()
            _lib.check(L.slak_mask_read_stats(self._plan, stats, self._stream()), "slak_mask_read_stats")
        self._bump_versions([t for _, t in params])
        self._nonzeros_after = {}
        for i, (name, _) in enumerate(params):
            self.name2nonzeros[name] = stats[4 * i + 0]
            self.name2zeros[name] = stats[4 * i + 1]
            self.name2removed[name] = stats[4 * i + 2]
            self._nonzeros_after[name] = stats[4 * i + 3]
        self._verify_ranks()

    # ------------------------------------------------------------------ utilities
    def get_momentum_for_weight(self, weight):
        """sparse_core.py:362-370 (read by the momentum growth / redistribution functions of funcs.py, which are not on this path;
        kept as the accessor it is)."""
        st = self.optimizer.state[weight]
        if 'exp_avg' in st:
            return st['exp_avg'] / (torch.sqrt(st['exp_avg_sq']) + 1e-08)
        if 'momentum_buffer' in st:
            return st['momentum_buffer']
        raise KeyError("optimizer state of this weight has neither 'exp_avg' nor 'momentum_buffer'")

    def get_gradient_for_weights(self, weight):
        return weight.grad.clone()

    def fired_masks_update(self):
        """sparse_core.py:388-402: which weights have ever been active.  (The reference never creates ``fired_masks``; here it
        starts as a copy of the current masks on first use.)"""
        if not hasattr(self, "fired_masks"):
            self.fired_masks = {}
        ntotal_fired_weights = ntotal_weights = 0.0
        layer_fired_weights = {}
        for module in self.modules:
            for name, weight in module.named_parameters():
                if name not in self.masks:
                    continue
                prev = self.fired_masks.get(name)
                cur = self.masks[name].data.byte()
                self.fired_masks[name] = cur if prev is None else (cur | prev.data.byte())
                fired = float(self.fired_masks[name].sum().item())
                ntotal_fired_weights += fired
                ntotal_weights += float(self.fired_masks[name].numel())
                layer_fired_weights[name] = fired / float(self.fired_masks[name].numel())
        total_fired_weights = ntotal_fired_weights / ntotal_weights
        print('The percentage of the total fired weights is:', total_fired_weights)
        return layer_fired_weights, total_fired_weights

    def print_nonzero_counts(self):
        after = getattr(self, "_nonzeros_after", {})
        for module in self.modules:
            for name, tensor in module.named_parameters():
                if name not in self.masks:
                    continue
                mask = self.masks[name]
                num_nonzeros = int(after[name]) if name in after else (mask != 0).sum().item()
                print('{0}: {1}->{2}, density: {3:.3f}'.format(name, self.name2nonzeros.get(name), num_nonzeros,
                                                               num_nonzeros / float(mask.numel())))
        print('Prune rate: {0}\n'.format(self.prune_rate))

    # ------------------------------------------------------------------ persistence
    _BITW = (128, 64, 32, 16, 8, 4, 2, 1)               # numpy.packbits bit order (most significant bit first)

    @classmethod
    def pack_mask(cls, m):
        """fp32 0/1 mask -> uint8 CPU tensor, 1 bit per weight, numpy.packbits layout."""
        bits = (m.reshape(-1) != 0).to(torch.int32)
        pad = (-bits.numel()) % 8
        if pad:
            bits = torch.cat([bits, bits.new_zeros(pad)])
        return (bits.view(-1, 8) * torch.tensor(cls._BITW, dtype=torch.int32, device=bits.device)).sum(1).to(torch.uint8).cpu()

    @classmethod
    def unpack_mask(cls, packed, shape):
        n = 1
        for d in shape:
            n *= d
        w = torch.tensor(cls._BITW, dtype=torch.int32)
        return ((packed.to(torch.int32).view(-1, 1) & w) != 0).reshape(-1)[:n].to(torch.float32).view(shape)

    def state_dict(self):
        """Everything a checkpoint needs to resume a sparse run with the SAME masks: the masks bit-packed (1 bit per weight, MSB
        first like numpy.packbits), the step counter, the current prune rate and the scheduler position.  The reference does not
        save masks: a resumed run re-derives them as ``weight != 0`` (``--sparse_init resume``, sparse_core.py:158-172), which
        drops the mask bit of every kept weight that happens to be exactly zero (e.g. freshly regrown ones: funcs.py:196-205
        grows weights at value 0) and forgets the prune-rate schedule.  utils.save_model (utils.py:447-469) has no mask entry;
        slak_amd.checkpoint.save_model adds this dict under ``'mask'``."""
        packed = {name: self.pack_mask(m) for name, m in self.masks.items()}
        shapes = {name: tuple(m.shape) for name, m in self.masks.items()}
        return {"masks": packed, "shapes": shapes, "steps": self.steps, "prune_rate": self.prune_rate,
                "decay_last_epoch": self.prune_rate_decay.cosine_stepper.last_epoch if hasattr(self.prune_rate_decay, "cosine_stepper") else None,
                "density": getattr(self, "density", None), "baseline_nonzero": self.baseline_nonzero}

    def load_state_dict(self, state):
        """Inverse of state_dict() on a Masking that went through add_module() on the same architecture: masks are overwritten IN
        PLACE (device plans and the optimizer's mask bindings stay valid), re-applied to the weights, and the schedule is put back."""
        shapes = state["shapes"]
        self._params_cache = None
        known = {}
        for module in self.modules:
            for name, tensor in module.named_parameters():
                known[name] = tensor
        # the keys carry the 'module.' prefix of whatever wrapper was passed to add_module (the reference registers the DDP
        # wrapper: main.py:425); a checkpoint written under DDP resumes into a bare model and vice versa
        def _resolve(saved):
            if saved in known:
                return saved
            alt = saved[len("module."):] if saved.startswith("module.") else "module." + saved
            return alt if alt in known else None
        renamed_masks, renamed_shapes = {}, {}
        for name in state["masks"]:
            tgt = _resolve(name)
            if tgt is None or tuple(known[tgt].shape) != tuple(shapes[name]):
                raise KeyError("saved mask %r does not match a parameter of the registered modules" % name)
            renamed_masks[tgt] = state["masks"][name]; renamed_shapes[tgt] = shapes[name]
        shapes = renamed_shapes
        new = {}
        for name, packed in renamed_masks.items():
            bits = self.unpack_mask(packed, shapes[name])
            if name in self.masks:
                self.masks[name].copy_(bits)
                new[name] = self.masks[name]
            else:
                new[name] = bits.to(self.device)
        self.masks = new                                  # entries the saved run had popped as dense stay unmasked
        self.names = list(new.keys())
        self.steps = int(state["steps"])
        self.prune_rate = state["prune_rate"]
        if state.get("density") is not None:
            self.density = state["density"]
        self.baseline_nonzero = state.get("baseline_nonzero", self.baseline_nonzero)
        last = state.get("decay_last_epoch")
        if last is not None and hasattr(self.prune_rate_decay, "cosine_stepper"):
            st = self.prune_rate_decay.cosine_stepper
            while st.last_epoch < last:                  # closed-form schedule: replaying the steps reproduces the fp64 rate exactly
                st.step()
        self._plan_key = None
        self._opt_masks_key = None
        self._synced_once = False
        self.apply_mask()

    def mask_checksum(self):
        """64-bit order-independent checksum of all masks (device kernel); equal on every rank iff masks agree."""
        self._ensure_plan()
        out = ctypes.c_ulonglong(0)
        with _on(self.device):
            _lib.check(_lib.lib().slak_mask_checksum(self._plan, ctypes.byref(out), self._stream()), "slak_mask_checksum")
        return out.value

    def ranks_agree(self):
        """True iff every rank holds the same masks (64-bit checksum, one 24-byte MIN and MAX all-reduce)."""
        import torch.distributed as dist
        c = self.mask_checksum()
        t = torch.tensor([c & 0x7fffffff, (c >> 31) & 0x7fffffff, c >> 62], dtype=torch.int64, device=self.device)
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(torch.equal(lo, hi))

    def check_rank_agreement(self):
        if not self.ranks_agree():
            raise RuntimeError("mask checksum differs across ranks")

    def _verify_ranks(self):
        """After every prune-and-grow round and load_state_dict of a distributed run (see __init__)."""
        import torch.distributed as dist
        if not (self.args.distributed and (self.check_ranks or self.debug_check_ranks)):
            return
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        if self.ranks_agree():
            return
        if self.debug_check_ranks:
            raise RuntimeError("mask checksum differs across ranks")
        self.rank_resyncs += 1
        print("slak_amd.Masking: masks diverged across ranks (resync #%d): re-broadcasting rank 0's masks" % self.rank_resyncs)
        self._synced_once = False
        self.apply_mask()                                 # synchronism_masks() + w *= mask, as every reference step does

    def synchronism_masks(self):
        """The reference broadcasts every mask from rank 0 on EVERY apply_mask (sparse_core.py:404-407: 95
        blocking collectives per step for SLaK-T).  Rank-0-wins is only observable once: masks come from
        per-rank RNG at init (main.py:232), and from then on they are a deterministic function of replicated
        weights and all-reduced gradients, so every rank computes the same masks.  One coalesced broadcast
        after (re-)initialisation preserves the reference's semantics; later calls are free."""
        if self._synced_once:
            return
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            names = list(self.masks.keys())
            flat = torch.cat([self.masks[n].reshape(-1) for n in names])
            dist.broadcast(flat, src=0)
            off = 0
            for n in names:
                k = self.masks[n].numel()
                self.masks[n].copy_(flat[off:off + k].view_as(self.masks[n]))
                off += k
        self._synced_once = True

    def __del__(self):
        try:
            if self._plan is not None:
                _lib.lib().slak_mask_plan_destroy(self._plan)
        except Exception:
            pass
