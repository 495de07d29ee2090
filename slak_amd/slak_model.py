"""Caller-side mirror of the reference model (models/SLaK.py) -- the code that CALLS the hot path.

It exists for two reasons only: (i) bench.py needs the SLaK-T 51x51 train step BASELINE.json names, and
/root/reference is not present on the GPU box; (ii) the full-model parity tests need the exact parameter
set (95 maskable tensors / 30,816,232 params for SLaK-T).  Module names, parameter shapes and state-dict keys
are identical to the reference (``stages.{i}.{j}.large_kernel.{LoRA1,LoRA2,small_conv}.{conv,bn}.*``,
``downsample_layers``, ``norm``, ``head``: models/SLaK.py:186-215), so checkpoints interchange.  timm is not
installed here; the two helpers it provided are restated (DropPath: timm1/layers/drop.py:137-166 semantics,
trunc_normal_: absolute bounds [-2, 2], timm1/layers/weight_init.py:43-67 == torch.nn.init.trunc_normal_).

One deliberate addition, off by default: ``lowp_dwconv=True`` makes the large-kernel block hand its input to
the three depthwise convs in the autocast dtype (bf16).  In the reference the residual stream reaching the op
is fp32 even under autocast (gamma is fp32: models/SLaK.py:149, :161-162) and custom_fwd casts to fp32
(depthwise_conv2d_implicit_gemm.py:16), so its "AMP" run still executes the fp32 SIMT kernels.
BASELINE.json's metric is a bf16 train step, so bench.py turns this on; parity tests cover both settings.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .depthwise_conv2d_implicit_gemm import DepthWiseConv2dImplicitGEMM

use_sync_bn = True       # same module-level switch as models/SLaK.py:19


def get_conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, bias):
    # the reference ignores everything but channels/kernel/bias as well (models/SLaK.py:21-22)
    return DepthWiseConv2dImplicitGEMM(in_channels, kernel_size, bias=bias)


def get_bn(channels):
    return nn.SyncBatchNorm(channels) if use_sync_bn else nn.BatchNorm2d(channels)


def conv_bn(in_channels, out_channels, kernel_size, stride, padding, groups, dilation=1, bn=True):
    seq = nn.Sequential()
    seq.add_module('conv', get_conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, False))
    if bn:
        seq.add_module('bn', get_bn(out_channels))
    return seq


def conv_bn_relu(in_channels, out_channels, kernel_size, stride, padding, groups, dilation=1):
    """conv -> BN -> ReLU (models/SLaK.py:30-36; not used by the SLaK blocks themselves, kept for the module surface)."""
    seq = conv_bn(in_channels, out_channels, kernel_size, stride, kernel_size // 2 if padding is None else padding, groups, dilation)
    seq.add_module('nonlinear', nn.ReLU())
    return seq


def fuse_bn(conv, bn):
    std = (bn.running_var + bn.eps).sqrt()
    scale = (bn.weight / std).reshape(-1, 1, 1, 1)
    return conv.weight * scale, bn.bias - bn.running_mean * bn.weight / std


class DropPath(nn.Module):
    """Stochastic depth per sample."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        if keep > 0.0:
            mask.div_(keep)
        return x * mask


class LayerNorm(nn.Module):
    """channels_last -> F.layer_norm; channels_first -> explicit mean/var over dim 1 (models/SLaK.py:235-261)."""

    def __init__(self, normalized_shape, eps=1e-6, data_format="channels_last"):
        super().__init__()
        if data_format not in ("channels_last", "channels_first"):
            raise NotImplementedError
        self.weight = nn.Parameter(torch.ones(normalized_shape))
        self.bias = nn.Parameter(torch.zeros(normalized_shape))
        self.eps = eps
        self.data_format = data_format
        self.normalized_shape = (normalized_shape,)

    def forward(self, x):
        if self.data_format == "channels_last":
            return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
        if self.fused_cf and x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and torch.is_autocast_enabled():
            # same arithmetic (fp32 statistics, fp32 result) as the explicit ops below, one HIP kernel per direction
            from . import block_ops
            return block_ops.ln_channels_first(x.contiguous(), self.weight, self.bias, self.eps, torch.float32)
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        x = (x - u) / torch.sqrt(s + self.eps)
        return self.weight[:, None, None] * x + self.bias[:, None, None]


class ReparamLargeKernelConv(nn.Module):
    """K x small + small x K (+ small x small) depthwise branches, each conv -> BN, summed (models/SLaK.py:60-100)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, groups, small_kernel,
                 small_kernel_merged=False, Decom=False, bn=True, lowp_dwconv=False):
        super().__init__()
        self.kernel_size = kernel_size
        self.small_kernel = small_kernel
        self.Decom = Decom
        self.lowp_dwconv = lowp_dwconv
        padding = kernel_size // 2
        if small_kernel_merged:
            self.lkb_reparam = get_conv2d(in_channels, out_channels, kernel_size, stride, padding, 1, groups, True)
            return
        if Decom:
            self.LoRA1 = conv_bn(in_channels, out_channels, (kernel_size, small_kernel), stride, padding, groups, bn=bn)
            self.LoRA2 = conv_bn(in_channels, out_channels, (small_kernel, kernel_size), stride, padding, groups, bn=bn)
        else:
            self.lkb_origin = conv_bn(in_channels, out_channels, kernel_size, stride, padding, groups, bn=bn)
        if small_kernel is not None and small_kernel < kernel_size:
            self.small_conv = conv_bn(in_channels, out_channels, small_kernel, stride, small_kernel // 2, groups, bn=bn)

    def forward(self, inputs, lowp=None):
        """``lowp``: the copy of ``inputs`` in the autocast dtype that the previous block's fused tail wrote alongside its fp32
        output (Block.emit_lowp) -- an explicit autograd output of that block, handed over by Block.forward."""
        if self.lowp_dwconv and torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            inputs = lowp if (lowp is not None and lowp.dtype == dt and lowp.shape == inputs.shape) else inputs.to(dt)
        if hasattr(self, 'lkb_reparam'):
            return self.lkb_reparam(inputs)
        if hasattr(self, 'reparam_bias'):                         # merge_kernel() on the decomposed path: thin kernels with the BNs folded in
            convs = [getattr(self, n) for n in ('LoRA1_reparam', 'LoRA2_reparam', 'small_conv_reparam') if hasattr(self, n)]
            if (len(convs) == 3 and inputs.is_cuda and inputs.dtype in (torch.bfloat16, torch.float16) and not torch.is_grad_enabled()
                    and tuple(convs[2].kernel_size) == (5, 5) and convs[0].kernel_size[1] == 5 and convs[1].kernel_size[0] == 5):
                from . import block_ops                           # one launch where the tri kernel covers the shape
                return block_ops.tri_dwconv_sum(inputs.contiguous(), convs[0].weight, convs[1].weight, convs[2].weight, self.reparam_bias)
            out = convs[0](inputs)
            for c in convs[1:]:
                out = out + c(inputs)
            return out + self.reparam_bias.to(out.dtype).view(1, -1, 1, 1)
        if (self.fused_bn and self.Decom and hasattr(self, 'small_conv') and inputs.is_cuda and inputs.dtype == torch.bfloat16
                and hasattr(self.LoRA1, 'bn') and (self.training or not torch.is_grad_enabled())
                and all(m.bn.track_running_stats and m.bn.running_mean is not None and m.bn.affine
                        for m in (self.LoRA1, self.LoRA2, self.small_conv))):      # what the fused op reads and updates
            # same arithmetic, the three BatchNorms and the two adds as one HIP op (slak_amd/block_ops.py)
            from . import block_ops
            c1, c2, c3 = self.LoRA1.conv, self.LoRA2.conv, self.small_conv.conv
            if (self.fused_tri and c1.bias is None and c2.bias is None and c3.bias is None and tuple(c3.kernel_size) == (5, 5)
                    and c1.kernel_size[1] == 5 and c2.kernel_size[0] == 5 and c1.kernel_size[0] == c2.kernel_size[1] and c1.kernel_size[0] > 5):
                # one autograd node for the three branches; in training its forward launch also leaves the BatchNorms' batch sums
                y1, y2, y3, st = block_ops.tri_dwconv(inputs.contiguous(), c1.weight, c2.weight, c3.weight, want_stats=2 if self.training else 1)
                return block_ops.branch_bn3(y1, y2, y3, self.LoRA1.bn, self.LoRA2.bn, self.small_conv.bn, stats=st)
            y1, y2, y3 = c1(inputs), c2(inputs), c3(inputs)
            return block_ops.branch_bn3(y1, y2, y3, self.LoRA1.bn, self.LoRA2.bn, self.small_conv.bn)
        if self.Decom:
            out = self.LoRA1(inputs) + self.LoRA2(inputs)
        else:
            out = self.lkb_origin(inputs)
        if hasattr(self, 'small_conv'):
            out += self.small_conv(inputs)
        return out

    def get_equivalent_kernel_bias(self):
        k, b = fuse_bn(self.lkb_origin.conv, self.lkb_origin.bn)
        if hasattr(self, 'small_conv'):
            sk, sb = fuse_bn(self.small_conv.conv, self.small_conv.bn)
            b = b + sb
            k = k + F.pad(sk, [(self.kernel_size - self.small_kernel) // 2] * 4)
        return k, b

    def merge_kernel(self):
        """Inference re-parameterisation (models/SLaK.py:112-122).  The reference can only merge the dense ``lkb_origin`` path; for
        the decomposed path the BatchNorm of every branch is folded into its thin kernel (fuse_bn, models/SLaK.py:51-58)
        and the three biases into one: ``LoRA1_reparam`` / ``LoRA2_reparam`` / ``small_conv_reparam`` (bias-free convs) +
        ``reparam_bias``.  The K x 5 + 5 x K + 5 x 5 taps are kept as three thin kernels (535 taps at K = 51) rather than scattered
        into one dense K x K kernel (2601 taps): the merged block runs as one launch of the three-branch kernel."""
        if self.Decom:
            names = ['LoRA1', 'LoRA2'] + (['small_conv'] if hasattr(self, 'small_conv') else [])
            bias = None
            for n in names:
                br = getattr(self, n)
                if hasattr(br, 'bn'):
                    k, b = fuse_bn(br.conv, br.bn)
                else:
                    k, b = br.conv.weight, torch.zeros_like(br.conv.weight[:, 0, 0, 0])
                c = br.conv
                merged = get_conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, c.dilation, c.groups, False)
                merged.weight.data = k.detach().clone()
                setattr(self, n + '_reparam', merged)
                bias = b.detach().clone() if bias is None else bias + b.detach()
                delattr(self, n)
            self.reparam_bias = nn.Parameter(bias)
            return
        k, b = self.get_equivalent_kernel_bias()
        c = self.lkb_origin.conv
        self.lkb_reparam = get_conv2d(c.in_channels, c.out_channels, c.kernel_size, c.stride, c.padding, c.dilation, c.groups, True)
        self.lkb_reparam.weight.data = k
        self.lkb_reparam.bias.data = b
        del self.lkb_origin
        if hasattr(self, 'small_conv'):
            del self.small_conv


class Block(nn.Module):
    """dw large-kernel -> LN (channels_last) -> Linear(C,4C) -> GELU -> Linear(4C,C) -> gamma -> residual."""

    def __init__(self, dim, drop_path=0., layer_scale_init_value=1e-6, kernel_size=(7, 7), Decom=None, bn=True, lowp_dwconv=False):
        super().__init__()
        self.large_kernel = ReparamLargeKernelConv(dim, dim, kernel_size[0], stride=1, groups=dim, small_kernel=kernel_size[1],
                                                   small_kernel_merged=False, Decom=Decom, bn=bn, lowp_dwconv=lowp_dwconv)
        self.norm = LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.gamma = nn.Parameter(layer_scale_init_value * torch.ones((dim)), requires_grad=True) if layer_scale_init_value > 0 else None
        self.drop_path = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def forward(self, x):
        """Tensor in, Tensor out -- what hooks, FLOPs counters and feature extractors see (models/SLaK.py:153-166)."""
        return self.forward_pair(x, None)[0]

    def forward_pair(self, x, x_lowp=None):
        """(out, out_lowp): ``out_lowp`` is the copy of ``out`` in the autocast dtype that the fused tail writes alongside its fp32 output
        when ``emit_lowp`` is set (None otherwise), ``x_lowp`` the previous block's.  Only ``_Stage.forward`` threads the pair from block
        to block; it never leaves a stage."""
        if self.fused_block and self.training and x.is_cuda and torch.is_grad_enabled():
            r = self._forward_fused_block(x, x_lowp)
            if r is not None:
                return r
        shortcut = x
        x = self.large_kernel(x, lowp=x_lowp)
        # the fused tail kernels take even C <= 1024 (tail_args_ok in csrc/block_tail.hip); anything else runs the PyTorch ops below
        if (self.fused_tail and x.is_cuda and x.dtype == torch.bfloat16 and self.gamma is not None
                and x.shape[1] % 2 == 0 and x.shape[1] <= 1024):
            r = self._forward_fused_tail(shortcut, x)
            return r if isinstance(r, tuple) else (r, None)
        self.__dict__.pop("_pending_scale", None)               # drawn for the fused tail only
        x = x.permute(0, 2, 3, 1)
        x = self.pwconv2(self.act(self.pwconv1(self.norm(x))))
        if self.gamma is not None:
            x = self.gamma * x
        x = x.permute(0, 3, 1, 2)
        return shortcut + self.drop_path(x), None


class _Stage(nn.Sequential):
    """A stage of blocks (models/SLaK.py:201-206 builds an nn.Sequential; this IS one: same state-dict keys, indexing, iteration).  Its
    forward hands each block's low-precision output copy to the next block explicitly, so every module boundary carries a Tensor.  A block
    with forward (pre-)hooks or backward (pre-)hooks is called through ``__call__`` like any module -- the hooks see tensors -- and simply gets
    no hand-off."""

    def forward(self, x, lowp=None):
        """``lowp``: a copy of ``x`` in the autocast dtype made by the producer of ``x`` (the stem's LayerNorm writes one: SLaK._forward_features),
        handed to the first block like a block's own hand-off."""
        import torch.nn.modules.module as _m
        # the fast-path test of nn.Module._call_impl: ANY hook -- forward, forward-pre, full backward, backward-pre, per module or global --
        # sends the block through __call__ (register_full_backward_hook on a Block fires as on any module)
        global_hooks = bool(_m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks or _m._global_backward_pre_hooks
                            or getattr(_m, "_global_forward_hooks_always_called", None))
        for blk in self:
            if isinstance(blk, Block) and not (global_hooks or blk._forward_hooks or blk._forward_pre_hooks or blk._backward_hooks
                                               or blk._backward_pre_hooks):
                x, lowp = blk.forward_pair(x, lowp)
            else:
                x, lowp = blk(x), None
        return x


def _block_forward_fused_tail(self, shortcut, x):
    """Same arithmetic as the lines below it in Block.forward, with the permute/LayerNorm and gamma/permute/residual steps as
    one HIP kernel each (slak_amd/block_ops.py); the two Linear layers and GELU are unchanged."""
    from . import block_ops
    t = block_ops.ln_nchw_to_nhwc(x.contiguous(), self.norm.weight.float(), self.norm.bias.float(), self.norm.eps)
    scale = None
    dp = self.drop_path
    if isinstance(dp, DropPath) and dp.drop_prob > 0.0 and self.training:
        scale = self.__dict__.pop("_pending_scale", None)        # drawn for all blocks at once by SLaK.forward_features
        if scale is None or scale.shape[0] != x.shape[0] or scale.device != x.device:
            keep = 1.0 - dp.drop_prob
            scale = torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep)
            if keep > 0.0:
                scale.div_(keep)
    emit = bool(self.emit_lowp and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)
    # MLP + gamma + permute + residual as one autograd node; with `emit` it returns (out, bf16 copy of out): both are autograd
    # outputs, the pair travels to the next block through _Stage.forward (Block.forward_pair)
    return block_ops.mlp_scale_residual(shortcut.contiguous(), t, self.pwconv1.weight, self.pwconv1.bias, self.pwconv2.weight,
                                        self.pwconv2.bias, self.gamma.float(), scale, emit_lowp=emit)


def _block_forward_fused_block(self, x, x_lowp):
    """The whole block as ONE autograd node (block_ops.fused_block: the same launches as the fused ops above, one Python call forward and one
    backward; under SyncBatchNorm the backward hides the statistics all-reduce behind the pointwise weight gradients).  None: a precondition of
    one of the fused ops does not hold -- the caller takes the ordinary path."""
    lk = self.large_kernel
    if not (self.fused_tail and lk.fused_bn and lk.fused_tri and lk.Decom and lk.lowp_dwconv and self.gamma is not None
            and hasattr(lk, "small_conv") and hasattr(lk, "LoRA1") and hasattr(lk.LoRA1, "bn")
            and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16) and x.shape[1] % 2 == 0 and x.shape[1] <= 1024):
        return None
    c1, c2, c3 = lk.LoRA1.conv, lk.LoRA2.conv, lk.small_conv.conv
    bns = (lk.LoRA1.bn, lk.LoRA2.bn, lk.small_conv.bn)
    if not (c1.bias is None and c2.bias is None and c3.bias is None and tuple(c3.kernel_size) == (5, 5) and c1.kernel_size[1] == 5
            and c2.kernel_size[0] == 5 and c1.kernel_size[0] == c2.kernel_size[1] and c1.kernel_size[0] > 5
            and all(b.training and b.track_running_stats and b.running_mean is not None and b.affine for b in bns)
            and all(w.dtype == torch.float32 for w in (c1.weight, c2.weight, c3.weight, self.norm.weight, self.gamma))):
        return None
    from . import block_ops
    if x_lowp is not None and not (x_lowp.dtype == torch.bfloat16 and x_lowp.shape == x.shape):
        x_lowp = None
    scale = None
    dp = self.drop_path
    if isinstance(dp, DropPath) and dp.drop_prob > 0.0:
        scale = self.__dict__.pop("_pending_scale", None)        # drawn for all blocks at once by SLaK.forward_features
        if scale is None or scale.shape[0] != x.shape[0] or scale.device != x.device:
            keep = 1.0 - dp.drop_prob
            scale = torch.empty(x.shape[0], device=x.device, dtype=torch.float32).bernoulli_(keep)
            if keep > 0.0:
                scale.div_(keep)
    r = block_ops.fused_block(x.contiguous(), x_lowp, c1.weight, c2.weight, c3.weight, bns, self.norm.weight, self.norm.bias, self.norm.eps,
                              self.pwconv1.weight, self.pwconv1.bias, self.pwconv2.weight, self.pwconv2.bias, self.gamma, scale,
                              emit_lowp=bool(self.emit_lowp))
    return r if isinstance(r, tuple) else (r, None)


Block._forward_fused_tail = _block_forward_fused_tail
Block._forward_fused_block = _block_forward_fused_block
Block.fused_block = False    # the whole block as one autograd node (needs fused_tail + the large kernel's fused_bn / fused_tri); bench.py turns it on
Block.fused_tail = False
Block.emit_lowp = False      # per instance: the next module is another Block with lowp_dwconv (set by SLaK.__init__ / bench.py)
ReparamLargeKernelConv.fused_bn = False
ReparamLargeKernelConv.fused_tri = False     # the three branch convs as one autograd node (block_ops.tri_dwconv); needs fused_bn
LayerNorm.fused_cf = False


class SLaK(nn.Module):
    def __init__(self, in_chans=3, num_classes=1000, depths=(3, 3, 9, 3), dims=(96, 192, 384, 768), drop_path_rate=0.,
                 layer_scale_init_value=1e-6, head_init_scale=1., kernel_size=(51, 49, 47, 13, 5), width_factor=1.0,
                 Decom=None, bn=True, lowp_dwconv=False):
        super().__init__()
        dims = [int(d * width_factor) for d in dims]
        self.kernel_size = list(kernel_size)
        self.downsample_layers = nn.ModuleList()
        self.downsample_layers.append(nn.Sequential(nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4),
                                                    LayerNorm(dims[0], eps=1e-6, data_format="channels_first")))
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(LayerNorm(dims[i], eps=1e-6, data_format="channels_first"),
                                                        nn.Conv2d(dims[i], dims[i + 1], kernel_size=2, stride=2)))
        rates = [r.item() for r in torch.linspace(0, drop_path_rate, sum(depths))]
        self.stages = nn.ModuleList()
        at = 0
        for i in range(4):
            self.stages.append(_Stage(*[
                Block(dims[i], drop_path=rates[at + j], layer_scale_init_value=layer_scale_init_value,
                      kernel_size=(self.kernel_size[i], self.kernel_size[-1]), Decom=Decom, bn=bn, lowp_dwconv=lowp_dwconv)
                for j in range(depths[i])]))
            at += depths[i]
            if lowp_dwconv:                                     # every block that hands its output to another block of the stage
                for blk in list(self.stages[-1])[:-1]:
                    blk.emit_lowp = True
        self.norm = nn.LayerNorm(dims[-1], eps=1e-6)
        self.head = nn.Linear(dims[-1], num_classes)
        self.apply(self._init_weights)
        self.head.weight.data.mul_(head_init_scale)
        self.head.bias.data.mul_(head_init_scale)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):           # DepthWiseConv2dImplicitGEMM IS an nn.Conv2d
            nn.init.trunc_normal_(m.weight, std=.02, a=-2., b=2.)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def structural_reparam(self):
        """Fold every block's branch BatchNorms for inference (ReparamLargeKernelConv.merge_kernel on each block)."""
        for m in self.modules():
            if isinstance(m, ReparamLargeKernelConv) and not hasattr(m, 'lkb_reparam') and not hasattr(m, 'reparam_bias'):
                m.merge_kernel()
        return self

    def _begin_counters(self):
        """One launch per training forward for all the step counters of the blocks' fused BatchNorms (block_ops.BnCounterPool)."""
        from . import block_ops
        pool = getattr(self, "_bn_pool", None)
        if pool is None or not pool.intact():
            bns = []
            for m in self.modules():
                if (isinstance(m, ReparamLargeKernelConv) and m.fused_bn and m.Decom and hasattr(m, "small_conv") and hasattr(m.LoRA1, "bn")):
                    trio = [m.LoRA1.bn, m.LoRA2.bn, m.small_conv.bn]
                    if all(b.track_running_stats and b.num_batches_tracked is not None and b.num_batches_tracked.is_cuda for b in trio):
                        bns += trio
            pool = block_ops.BnCounterPool(bns) if bns and len({b.num_batches_tracked.device for b in bns}) == 1 else None
            object.__setattr__(self, "_bn_pool", pool)           # (not a submodule / buffer: stays out of state_dict)
        if pool is not None:
            pool.begin_forward()

    def _draw_drop_path(self, x):
        """The per-sample stochastic-depth scales of every block in three launches (one uniform draw for all blocks) instead of two per
        block; same distribution as DropPath's bernoulli_(keep) / keep, handed to the fused block tails through Block._pending_scale."""
        blocks = [b for st in self.stages for b in st if isinstance(b, Block) and b.fused_tail
                  and isinstance(b.drop_path, DropPath) and 0.0 < b.drop_path.drop_prob < 1.0]
        if not blocks:
            return
        keeps = getattr(self, "_dp_keeps", None)
        if keeps is None or keeps.device != x.device or keeps.shape[0] != len(blocks):
            keeps = torch.tensor([1.0 - b.drop_path.drop_prob for b in blocks], dtype=torch.float32, device=x.device).view(-1, 1)
            object.__setattr__(self, "_dp_keeps", keeps)
        scales = (torch.rand(len(blocks), x.shape[0], device=x.device) < keeps).float() / keeps
        for i, b in enumerate(blocks):
            b.__dict__["_pending_scale"] = scales[i]

    def forward_features(self, x):
        managed = self.training and x.is_cuda
        if managed:
            self._begin_counters()
            self._draw_drop_path(x)
        try:
            return self._forward_features(x)
        finally:
            if managed:                                           # nothing of this forward outlives it: deepcopy / torch.save(model) stay clean
                pool = getattr(self, "_bn_pool", None)
                if pool is not None:
                    pool.end_forward()
                for st in self.stages:
                    for b in st:
                        b.__dict__.pop("_pending_scale", None)

    def _forward_features(self, x):
        import torch.nn.modules.module as _m
        # the fused stem / downsample paths below call kernels instead of the modules: with a GLOBAL module hook registered
        # (register_module_forward_hook & co.) every module has to be called, as _Stage.forward already decides for the blocks
        no_global_hooks = not (_m._global_forward_hooks or _m._global_forward_pre_hooks or _m._global_backward_hooks or _m._global_backward_pre_hooks
                               or getattr(_m, "_global_forward_hooks_always_called", None))
        for i in range(4):
            ds = self.downsample_layers[i]
            if (self.fused_downsample and no_global_hooks and i > 0 and x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled()
                    and isinstance(ds[0], LayerNorm) and ds[0].data_format == "channels_first" and isinstance(ds[1], nn.Conv2d)
                    and ds[1].kernel_size == (2, 2) and ds[1].stride == (2, 2) and ds[1].padding == (0, 0) and ds[1].groups == 1):
                from . import block_ops                           # LayerNorm + the 2x2 / stride-2 conv as one LN kernel + library GEMMs
                xc = x.contiguous()
                if block_ops.ln_patch_covers(xc):
                    x = self.stages[i](block_ops.downsample_ln_conv(xc, ds[0].weight, ds[0].bias, ds[1].weight, ds[1].bias, ds[0].eps))
                    continue
            if (self.fused_downsample and self.fused_stem and no_global_hooks and i == 0 and x.is_cuda and x.dtype == torch.float32 and torch.is_autocast_enabled() and not x.requires_grad
                    and isinstance(ds[0], nn.Conv2d) and ds[0].kernel_size == (4, 4) and ds[0].stride == (4, 4) and ds[0].padding == (0, 0)
                    and ds[0].groups == 1 and x.shape[2] % 4 == 0 and x.shape[3] % 4 == 0):
                from . import block_ops                           # stem conv as patch matrix + library GEMMs (no MIOpen launch, no layout transposes)
                y = block_ops.stem_conv(x.contiguous(), ds[0].weight, ds[0].bias)
                st = self.stages[i]
                if (self.stem_lowp_handoff and isinstance(ds[1], LayerNorm) and ds[1].data_format == "channels_first" and len(st) > 0
                        and isinstance(st[0], Block) and st[0].large_kernel.lowp_dwconv and torch.get_autocast_dtype("cuda") == torch.bfloat16
                        and not any(m._forward_hooks or m._forward_pre_hooks or m._backward_hooks or m._backward_pre_hooks for m in (ds[1], st))):
                    # the LayerNorm writes the bf16 copy the first block's convs read and takes that copy's gradient back (block_ops._LnChannelsFirstPair)
                    x = st(*block_ops.ln_channels_first_pair(y, ds[1].weight, ds[1].bias, ds[1].eps))
                else:
                    x = st(ds[1](y))
                continue
            x = self.stages[i](ds(x))
        return self.norm(x.mean([-2, -1]))

    def forward(self, x):
        return self.head(self.forward_features(x))


SLaK.stem_lowp_handoff = True                # (with fused_stem) the stem LayerNorm hands the first block its bf16 input copy (block_ops.ln_channels_first_pair)
SLaK.fused_stem = True                       # (with fused_downsample) the stem conv as patch matrix + library GEMMs (block_ops.stem_conv)
SLaK.fused_downsample = False                # downsample layers as LN-to-patch-matrix kernel + library GEMMs (block_ops.downsample_ln_conv)


_VARIANTS = {
    "tiny": dict(depths=(3, 3, 9, 3), dims=(96, 192, 384, 768)),
    "small": dict(depths=(3, 3, 27, 3), dims=(96, 192, 384, 768)),
    "base": dict(depths=(3, 3, 27, 3), dims=(128, 256, 512, 1024)),
    "large": dict(depths=(3, 3, 27, 3), dims=(192, 384, 768, 1536)),
}


def SLaK_tiny(pretrained=False, **kw): return SLaK(**_VARIANTS["tiny"], **kw)
def SLaK_small(pretrained=False, **kw): return SLaK(**_VARIANTS["small"], **kw)
def SLaK_base(pretrained=False, in_22k=False, **kw): return SLaK(**_VARIANTS["base"], **kw)
def SLaK_large(pretrained=False, in_22k=False, **kw): return SLaK(**_VARIANTS["large"], **kw)


def create_model(name, **kw):
    """Stand-in for timm's registry lookup used by main.py:301-312."""
    return {"SLaK_tiny": SLaK_tiny, "SLaK_small": SLaK_small, "SLaK_base": SLaK_base, "SLaK_large": SLaK_large}[name](**kw)


def slak_mask_set_shapes(variant="tiny", kernel_size=(51, 49, 47, 13, 5), num_classes=1000, in_chans=3, only_L=False):
    """Shapes of the tensors ``Masking.add_module`` would mask (2-D / 4-D parameters, sparse_core.py:121-130),
    computed arithmetically (no model instantiation)."""
    cfg = _VARIANTS[variant]
    dims, depths = cfg["dims"], cfg["depths"]
    # parameter order follows named_parameters(): downsample_layers first, then stages, then head
    shapes = []
    if not only_L:
        shapes.append((dims[0], in_chans, 4, 4))
        for i in range(3):
            shapes.append((dims[i + 1], dims[i], 2, 2))
    for i in range(4):
        K, s, C = kernel_size[i], kernel_size[-1], dims[i]
        for _ in range(depths[i]):
            shapes.append((C, 1, K, s)); shapes.append((C, 1, s, K))
            if not only_L:
                shapes.append((C, 1, s, s)); shapes.append((4 * C, C)); shapes.append((C, 4 * C))
    if not only_L:
        shapes.append((num_classes, dims[-1]))
    return shapes
