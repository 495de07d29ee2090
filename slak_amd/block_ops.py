"""Block-tail glue ops as autograd Functions over the C ABI (include/slak_hip.h, block_tail.hip).

    ln_nchw_to_nhwc(x, weight, bias, eps)            == F.layer_norm(x.permute(0,2,3,1), (C,), weight, bias, eps)   (bf16 out)
    scale_residual(shortcut, z, gamma, sample_scale) == shortcut + sample_scale[:,None,None,None] * (gamma * z).permute(0,3,1,2)

(models/SLaK.py:153-166, :253-255).  x bf16 NCHW, z bf16 NHWC; parameters, statistics and the residual stream fp32.
"""
import torch

from . import _lib
from .ops import _workspace, _stream


def _chk(t, name, dtype=None):
    if not t.is_cuda or not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous CUDA/HIP tensor" % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))


class _LnNchwToNhwc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _chk(x, "x", torch.bfloat16); _chk(weight, "weight", torch.float32); _chk(bias, "bias", torch.float32)
        N, C, H, W = x.shape
        y = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
        mean = torch.empty((N, H * W), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L = _lib.lib()
        with torch.cuda.device(x.device):
            _lib.check(L.slak_ln_nchw_to_nhwc_forward(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                      rstd.data_ptr(), N, C, H * W, float(eps), _stream(x.device)), "slak_ln_nchw_to_nhwc_forward")
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        g = g.contiguous()
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        dx = torch.empty_like(x)
        dw = torch.empty_like(weight); db = torch.empty_like(weight)
        L = _lib.lib()
        ws, nb = _workspace(L.slak_block_tail_workspace_bytes(N, C, H * W), x.device)
        with torch.cuda.device(x.device):
            _lib.check(L.slak_ln_nchw_to_nhwc_backward(g.data_ptr(), x.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                       dx.data_ptr(), dw.data_ptr(), db.data_ptr(), N, C, H * W,
                                                       ws.data_ptr(), nb, _stream(x.device)), "slak_ln_nchw_to_nhwc_backward")
        return dx, dw, db, None


class _ScaleResidual(torch.autograd.Function):
    @staticmethod
    def forward(ctx, shortcut, z, gamma, sample_scale):
        _chk(shortcut, "shortcut"); _chk(z, "z", torch.bfloat16); _chk(gamma, "gamma", torch.float32)
        N, C, H, W = shortcut.shape
        if z.shape != (N, H, W, C):
            raise RuntimeError("z must be (N,H,W,C)")
        sdt = {torch.float32: _lib.SLAK_F32, torch.bfloat16: _lib.SLAK_BF16}.get(shortcut.dtype)
        if sdt is None:
            raise TypeError("shortcut must be float32 or bfloat16")
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=z.device)
        L = _lib.lib()
        with torch.cuda.device(z.device):
            _lib.check(L.slak_scale_residual_forward(shortcut.data_ptr(), sdt, z.data_ptr(), gamma.data_ptr(),
                                                     sample_scale.data_ptr() if sample_scale is not None else None,
                                                     out.data_ptr(), N, C, H * W, _stream(z.device)), "slak_scale_residual_forward")
        ctx.save_for_backward(z, gamma, sample_scale)
        ctx.shortcut_dtype = shortcut.dtype
        return out

    @staticmethod
    def backward(ctx, dout):
        z, gamma, sample_scale = ctx.saved_tensors
        N, H, W, C = z.shape
        dout = dout.contiguous()
        if dout.dtype != torch.float32:
            dout = dout.float()
        dz = torch.empty_like(z)
        dgamma = torch.empty_like(gamma)
        L = _lib.lib()
        ws, nb = _workspace(L.slak_block_tail_workspace_bytes(N, C, H * W), z.device)
        with torch.cuda.device(z.device):
            _lib.check(L.slak_scale_residual_backward(dout.data_ptr(), z.data_ptr(), gamma.data_ptr(),
                                                      sample_scale.data_ptr() if sample_scale is not None else None,
                                                      dz.data_ptr(), dgamma.data_ptr(), N, C, H * W, ws.data_ptr(), nb, _stream(z.device)),
                       "slak_scale_residual_backward")
        dshortcut = dout if ctx.shortcut_dtype == torch.float32 else dout.to(ctx.shortcut_dtype)
        return dshortcut, dz, dgamma, None


def ln_nchw_to_nhwc(x, weight, bias, eps=1e-6):
    return _LnNchwToNhwc.apply(x, weight, bias, eps)


def scale_residual(shortcut, z, gamma, sample_scale=None):
    return _ScaleResidual.apply(shortcut, z, gamma, sample_scale)


class _LinearSplitK(torch.autograd.Function):
    """``F.linear`` under bf16 autocast with the weight gradient computed as a split-K batched GEMM.

    dW = dY^T X has K = N*H*W rows (401,408 for SLaK-T stage 1 at batch 128) against a 96x384 result; hipBLASLt's
    heuristic picks a single-CTA-per-tile kernel for it (34 TFLOP/s measured, 0.86 ms per call).  Viewing the rows as
    S slices and calling the library's batched GEMM gives S times the parallelism; the S partial products are added in fp32.
    Plain library GEMMs, no change in arithmetic: the forward and the data gradient are the usual calls."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        wb = weight.to(torch.bfloat16)
        y = torch.nn.functional.linear(x, wb, bias.to(torch.bfloat16) if bias is not None else None)
        ctx.save_for_backward(x, wb)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        x2 = x.reshape(-1, x.shape[-1]); dy2 = dy.reshape(-1, dy.shape[-1])
        dx = torch.mm(dy2, wb).view_as(x)
        M = x2.shape[0]
        S = max(1, M // 6272)
        while S > 1 and M % S:
            S -= 1
        if S > 1:
            dw = torch.bmm(dy2.view(S, M // S, -1).transpose(1, 2), x2.view(S, M // S, -1)).float().sum(0)
        else:
            dw = torch.mm(dy2.t(), x2).float()
        db = dy2.float().sum(0) if ctx.has_bias else None
        return dx, dw, db


def linear_splitk(x, weight, bias):
    return _LinearSplitK.apply(x, weight, bias)
