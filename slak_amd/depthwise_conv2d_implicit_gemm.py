"""Drop-in for the reference's ``depthwise_conv2d_implicit_gemm`` module
(depthwise_conv2d_implicit_gemm.py:11-66): same class name, constructor, parameter shape, state-dict
keys, dtype dispatch and error behaviour; the native module underneath is libslak_hip.so (gfx950 HIP).

Differences, all additive:
  * bf16 is supported (the reference raises TypeError for it: depthwise_conv2d_implicit_gemm.py:63);
  * kernels run on PyTorch's current stream (the reference uses the null stream: convolution.h:243);
  * failures raise instead of exit()ing the process (forward_fp32.cu:173-192);
  * OPT-IN (off by default, ``DepthWiseConv2dImplicitGEMM.fp32_matrix_cores_under_autocast = True`` or SLAK_FP32_AUTOCAST_SPLIT=1): an
    fp32 activation that reaches the op UNDER torch.autocast (the reference's default AMP flow: the residual stream is fp32 and
    depthwise_conv2d_implicit_gemm.py:16 keeps it so) runs on the bf16 matrix cores as a two-term split -- 16 significand bits per
    operand, fp32 accumulation and result: more precise than anything autocast does around it, 2.2x faster than the exact kernels, but
    not the reference's exact fp32 arithmetic.  By default fp32 follows the process-wide switch (ops.allow_fp32_matrix_cores /
    slak_set_fp32_matrix_cores, itself off by default): exact VALU kernels, in and out of autocast.
"""
import os

import torch
import torch.nn as nn

from . import ops as _extension

__all__ = ["DepthWiseConv2dImplicitGEMM"]

_NO_CTX = _extension._NULL


def _make_function(cast_dtype, name):
    # mirrors _DepthWiseConv2dImplicitGEMMFP32/FP16 (depthwise_conv2d_implicit_gemm.py:14-49).
    # fp32: custom_fwd(cast_inputs=float32) exactly as the reference (:16) -- under autocast an fp32 activation
    #       stays on the fp32 kernels.
    # fp16/bf16: the activation already has the low-precision dtype (that is how this branch was chosen);
    #       the fp32 master weight is deliberately NOT down-cast (the reference's cast_inputs=float16 rounds
    #       it, :35): the HIP kernels read fp32 weights directly and the fp32 dw (backward_filter_fp16.cu:187)
    #       reaches the fp32 parameter without a round trip through 16 bits.
    fwd_deco = (torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32) if cast_dtype == torch.float32
                else torch.amp.custom_fwd(device_type="cuda"))

    class _Fn(torch.autograd.Function):
        @staticmethod
        @fwd_deco
        def forward(ctx, x, w, split=False):
            ctx.save_for_backward(x, w)
            ctx.split = bool(split) and cast_dtype == torch.float32
            if ctx.split:
                with _extension.fp32_matrix_cores(True):
                    return _extension.dwconv2d_forward(x.contiguous(), w.contiguous())
            return _extension.dwconv2d_forward(x.contiguous(), w.contiguous())

        @staticmethod
        @torch.amp.custom_bwd(device_type="cuda")
        def backward(ctx, grad):
            x, w = ctx.saved_tensors
            grad = grad.contiguous()
            x = x.contiguous()
            w = w.contiguous()
            dx = dw = None
            with _extension.fp32_matrix_cores(True) if ctx.split else _NO_CTX:
                if ctx.needs_input_grad[0]:
                    dx = _extension.dwconv2d_backward_data(grad, w)
                if ctx.needs_input_grad[1]:
                    dw = _extension.dwconv2d_backward_filter(grad, x, w)      # fp32, like backward_filter_fp16.cu:187
                    if dw.dtype != w.dtype:
                        dw = dw.to(w.dtype)
            return dx, dw, None

    _Fn.__name__ = _Fn.__qualname__ = name
    return _Fn


_DepthWiseConv2dImplicitGEMMFP32 = _make_function(torch.float32, "_DepthWiseConv2dImplicitGEMMFP32")
_DepthWiseConv2dImplicitGEMMFP16 = _make_function(torch.float16, "_DepthWiseConv2dImplicitGEMMFP16")
_DepthWiseConv2dImplicitGEMMBF16 = _make_function(torch.bfloat16, "_DepthWiseConv2dImplicitGEMMBF16")


class DepthWiseConv2dImplicitGEMM(nn.Conv2d):
    """``nn.Conv2d`` subclass with weight ``(C,1,kh,kw)``; computes a stride-1 "same" depthwise conv
    regardless of ``self.padding`` (which stays (0,0), as in the reference)."""

    fp32_matrix_cores_under_autocast = os.environ.get("SLAK_FP32_AUTOCAST_SPLIT", "0") == "1"      # opt-in, like allow_tf32

    def __init__(self, channels, kernel, bias=False):
        super().__init__(channels, channels, kernel, groups=channels, bias=bias)

    def forward(self, x):
        if x.dtype == torch.float32:
            split = self.fp32_matrix_cores_under_autocast and x.is_cuda and torch.is_autocast_enabled("cuda")
            x = _DepthWiseConv2dImplicitGEMMFP32.apply(x, self.weight, split)
        elif x.dtype == torch.float16:
            x = _DepthWiseConv2dImplicitGEMMFP16.apply(x, self.weight, False)
        elif x.dtype == torch.bfloat16:
            x = _DepthWiseConv2dImplicitGEMMBF16.apply(x, self.weight, False)
        else:
            raise TypeError("Only support fp32, fp16 and bf16, get {}".format(x.dtype))
        if self.bias is not None:
            x = x + self.bias.to(x).view(1, -1, 1, 1)
        return x
