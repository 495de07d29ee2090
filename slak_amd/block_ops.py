"""Block-tail glue ops as autograd Functions over the C ABI (include/slak_hip.h, block_tail.hip).

    ln_nchw_to_nhwc(x, weight, bias, eps)            == F.layer_norm(x.permute(0,2,3,1), (C,), weight, bias, eps)   (bf16 out)
    scale_residual(shortcut, z, gamma, sample_scale) == shortcut + sample_scale[:,None,None,None] * (gamma * z).permute(0,3,1,2)

(models/SLaK.py:153-166, :253-255).  x bf16 NCHW, z bf16 NHWC; parameters, statistics and the residual stream fp32.
"""
import ctypes
import os

import torch

from . import _lib
from .ops import _workspace, _stream, _on


def _chk(t, name, dtype=None):
    if not t.is_cuda or not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous CUDA/HIP tensor" % name)
    if dtype is not None and t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))


def _ln_fwd_impl(x, weight, bias, eps):
    _chk(x, "x", torch.bfloat16); _chk(weight, "weight", torch.float32); _chk(bias, "bias", torch.float32)
    N, C, H, W = x.shape
    y = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
    mean = torch.empty((N, H * W), dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    L = _lib.lib()
    with _on(x.device):
        _lib.check(L.slak_ln_nchw_to_nhwc_forward(x.data_ptr(), weight.data_ptr(), bias.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                                  rstd.data_ptr(), N, C, H * W, float(eps), _stream(x.device)), "slak_ln_nchw_to_nhwc_forward")
    return y, mean, rstd


_tail_ws_cache = {}


def _tail_ws_bytes(N, C, P):
    key = (N, C, P)
    v = _tail_ws_cache.get(key)
    if v is None:
        v = _tail_ws_cache[key] = int(_lib.lib().slak_block_tail_workspace_bytes(N, C, P))
    return v


def _ln_bwd_impl(g, x, weight, mean, rstd):
    N, C, H, W = x.shape
    g = g.contiguous()
    if g.dtype != torch.bfloat16:
        g = g.to(torch.bfloat16)
    dx = torch.empty_like(x)
    dw = torch.empty_like(weight); db = torch.empty_like(weight)
    L = _lib.lib()
    ws, nb = _workspace(_tail_ws_bytes(N, C, H * W), x.device)
    with _on(x.device):
        _lib.check(L.slak_ln_nchw_to_nhwc_backward(g.data_ptr(), x.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                   dx.data_ptr(), dw.data_ptr(), db.data_ptr(), N, C, H * W,
                                                   ws.data_ptr() if ws is not None else None, nb, _stream(x.device)), "slak_ln_nchw_to_nhwc_backward")
    return dx, dw, db


class _LnNchwToNhwc(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y, mean, rstd = _ln_fwd_impl(x, weight, bias, eps)
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, rstd = ctx.saved_tensors
        dx, dw, db = _ln_bwd_impl(g, x, weight, mean, rstd)
        return dx, dw, db, None


def _scale_residual_fwd(shortcut, z, gamma, sample_scale, emit_lowp):
    _chk(shortcut, "shortcut"); _chk(z, "z", torch.bfloat16); _chk(gamma, "gamma", torch.float32)
    N, C, H, W = shortcut.shape
    if z.shape != (N, H, W, C):
        raise RuntimeError("z must be (N,H,W,C)")
    sdt = {torch.float32: _lib.SLAK_F32, torch.bfloat16: _lib.SLAK_BF16}.get(shortcut.dtype)
    if sdt is None:
        raise TypeError("shortcut must be float32 or bfloat16")
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=z.device)
    out16 = torch.empty((N, C, H, W), dtype=torch.bfloat16, device=z.device) if emit_lowp else None
    L = _lib.lib()
    with _on(z.device):
        _lib.check(L.slak_scale_residual_forward(shortcut.data_ptr(), sdt, z.data_ptr(), gamma.data_ptr(),
                                                 sample_scale.data_ptr() if sample_scale is not None else None,
                                                 out.data_ptr(), out16.data_ptr() if emit_lowp else None,
                                                 N, C, H * W, _stream(z.device)), "slak_scale_residual_forward")
    return out, out16


def _scale_residual_bwd(z, gamma, sample_scale, shortcut_dtype, dout, dout16):
    """-> (dshortcut, dz, dgamma, colsum of dz over (n, p) = the bias gradient of the Linear that produced z)"""
    N, H, W, C = z.shape
    if dout is None:                                   # only the bf16 copy was used downstream
        dout = torch.zeros((N, C, H, W), dtype=torch.float32, device=z.device)
    dout = dout.contiguous()
    if dout.dtype != torch.float32:
        dout = dout.float()
    if dout16 is not None:
        dout16 = dout16.contiguous()
        if dout16.dtype != torch.bfloat16:
            dout16 = dout16.to(torch.bfloat16)
    dsum = torch.empty_like(dout) if dout16 is not None else None
    dz = torch.empty_like(z)
    dgamma = torch.empty_like(gamma)
    dzc = torch.empty_like(gamma)
    L = _lib.lib()
    ws, nb = _workspace(L.slak_block_tail_workspace_bytes(N, C, H * W), z.device)
    with _on(z.device):
        _lib.check(L.slak_scale_residual_backward(dout.data_ptr(), dout16.data_ptr() if dout16 is not None else None,
                                                  dsum.data_ptr() if dsum is not None else None, z.data_ptr(), gamma.data_ptr(),
                                                  sample_scale.data_ptr() if sample_scale is not None else None,
                                                  dz.data_ptr(), dgamma.data_ptr(), dzc.data_ptr(), N, C, H * W,
                                                  ws.data_ptr() if ws is not None else None, nb, _stream(z.device)),
                   "slak_scale_residual_backward")
    dsc = dsum if dsum is not None else dout
    dshortcut = dsc if shortcut_dtype == torch.float32 else dsc.to(shortcut_dtype)
    return dshortcut, dz, dgamma, dzc


class _ScaleResidual(torch.autograd.Function):
    """out = shortcut + scale * gamma * z (NHWC -> NCHW), optionally with a second, bf16-rounded copy of `out` (the next block's
    depthwise-conv input under autocast: saves the fp32 -> bf16 cast pass forward and a cast + add pass backward)."""

    @staticmethod
    def forward(ctx, shortcut, z, gamma, sample_scale, emit_lowp):
        out, out16 = _scale_residual_fwd(shortcut, z, gamma, sample_scale, emit_lowp)
        ctx.save_for_backward(z, gamma, sample_scale)
        ctx.shortcut_dtype = shortcut.dtype
        return (out, out16) if emit_lowp else out

    @staticmethod
    def backward(ctx, dout, dout16=None):
        z, gamma, sample_scale = ctx.saved_tensors
        dshortcut, dz, dgamma, _ = _scale_residual_bwd(z, gamma, sample_scale, ctx.shortcut_dtype, dout, dout16)
        return dshortcut, dz, dgamma, None, None


accumulate_dgrad = os.environ.get("SLAK_DGRAD_ACC", "1") != "0"     # A/B switch: 0 = three plain launches + two tensor adds
fused_tri_wgrad = os.environ.get("SLAK_TRI_WGRAD", "1") != "0"      # A/B switch: 0 = three weight-gradient launches per block everywhere
fused_tri_backward = os.environ.get("SLAK_TRI_BACKWARD", "1") != "0"   # A/B switch: 0 = the data-gradient launch and the weight-gradient launch apart on the 14 x 14 class as well


_tri_plan_cache = {}


def _tri_plan(dt, N, C, H, W, K):
    """What the library answers for a block shape -- (one-launch forward?, one-launch data gradient?, rows of the forward launch's BatchNorm
    statistics, workspace bytes of the three-branch weight gradient, of the two-branch one, one-launch backward?) -- asked once per (dtype,
    shape): six ctypes calls per block and direction otherwise."""
    key = (dt, N, C, H, W, K)
    plan = _tri_plan_cache.get(key)
    if plan is None:
        L = _lib.lib()
        plan = (L.slak_dwconv2d_tri_supported_op(dt, N, C, H, W, K, 0) == 1, L.slak_dwconv2d_tri_supported_op(dt, N, C, H, W, K, 1) == 1,
                int(L.slak_dwconv2d_tri_stats_rows(dt, N, C, H, W, K)), int(L.slak_dwconv2d_tri_filter_workspace_bytes(dt, N, C, H, W, K)),
                int(L.slak_dwconv2d_pair_filter_workspace_bytes(dt, N, C, H, W, K)),
                L.slak_dwconv2d_tri_backward_supported(dt, N, C, H, W, K) == 1)      # [5]: data gradient + the three weight gradients in one launch
        _tri_plan_cache[key] = plan
    return plan


def _tri_forward_impl(x, wv, wh, ws, want_stats):
    """-> (yv, yh, ys, stats, tri_dgrad): the three branch outputs, the BatchNorm sums the launch(es) gathered (one [rows][C][6] array, a triple
    of [rows_b][C][2] arrays, or None) and whether the one-launch data gradient exists for this shape."""
    from . import ops
    _chk(x, "input")
    N, C, H, W = x.shape
    K = wv.shape[2]
    if wv.shape != (C, 1, K, 5) or wh.shape != (C, 1, 5, K) or ws.shape != (C, 1, 5, 5):
        raise RuntimeError("tri_dwconv expects filters (C,1,K,5), (C,1,5,K), (C,1,5,5)")
    L = _lib.lib()
    dt = ops._DT.get(x.dtype)
    # one launch for the three branches where the library says it wins (slak_dwconv2d_tri_supported_op: per op)
    f32w = all(w.dtype == torch.float32 and w.is_contiguous() for w in (wv, wh, ws))
    plan = _tri_plan(dt, N, C, H, W, K) if (dt is not None and f32w) else (False, False, 0, 0, 0, False)
    tri, tri_dgrad = plan[0], plan[1]
    stats = None
    if tri:
        yv, yh, ys = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        rows = plan[2] if (want_stats == 2 and bn_stats_in_conv) else 0
        with _on(x.device):
            if rows > 0:                                     # the launch also leaves the branch BatchNorms' batch statistics
                stats = torch.empty((rows, C, 6), dtype=torch.float32, device=x.device)
                _lib.check(L.slak_dwconv2d_tri_forward_stats(x.data_ptr(), wv.data_ptr(), wh.data_ptr(), ws.data_ptr(), yv.data_ptr(),
                                                             yh.data_ptr(), ys.data_ptr(), stats.data_ptr(), dt, N, C, H, W, K, _stream(x.device)),
                           "slak_dwconv2d_tri_forward_stats")
            else:
                _lib.check(L.slak_dwconv2d_tri_forward(x.data_ptr(), wv.data_ptr(), wh.data_ptr(), ws.data_ptr(), yv.data_ptr(),
                                                       yh.data_ptr(), ys.data_ptr(), dt, N, C, H, W, K, _stream(x.device)),
                           "slak_dwconv2d_tri_forward")
    elif want_stats == 2 and bn_stats_in_conv:               # per-branch launches that gather their BatchNorm's sums in the copy-out
        (yv, sv), (yh, sh), (ys, ss) = ops.dwconv2d_forward_stats(x, wv), ops.dwconv2d_forward_stats(x, wh), ops.dwconv2d_forward_stats(x, ws)
        if sv is not None and sh is not None and ss is not None:
            stats = (sv, sh, ss)
    else:
        yv, yh, ys = ops.dwconv2d_forward(x, wv), ops.dwconv2d_forward(x, wh), ops.dwconv2d_forward(x, ws)
    return yv, yh, ys, stats, tri_dgrad


def _tri_backward_impl(x, wv, wh, ws, dyv, dyh, dys, tri_dgrad, need_dx, need_w):
    """-> (dx, dwv, dwh, dws); need_w: three booleans."""
    from . import ops
    N, C, H, W = x.shape
    K = wv.shape[2]
    dyv, dyh, dys = (torch.zeros_like(x) if g is None else g for g in (dyv, dyh, dys))       # (a branch output nobody used)
    dyv, dyh, dys = (g.contiguous() if g.dtype == x.dtype else g.to(x.dtype).contiguous() for g in (dyv, dyh, dys))
    dx = None
    plan = _tri_plan(ops._DT[x.dtype], N, C, H, W, K) if x.dtype in ops._DT else (False, False, 0, 0, 0, False)
    if need_dx and all(need_w) and fused_tri_wgrad and fused_tri_backward and plan[5] and plan[3] and \
            all(w.dtype == torch.float32 and w.is_contiguous() for w in (wv, wh, ws)):
        # the whole backward in ONE launch (14 x 14 class): the dY planes are staged once for dx and for the three weight gradients
        L = _lib.lib()
        dx = torch.empty_like(x)
        dwv, dwh, dws = (torch.empty_like(w, dtype=torch.float32) for w in (wv, wh, ws))
        wsb, nbb = _workspace(plan[3], x.device)
        with _on(x.device):
            rc = L.slak_dwconv2d_tri_backward(dyv.data_ptr(), dyh.data_ptr(), dys.data_ptr(), x.data_ptr(), wv.data_ptr(), wh.data_ptr(), ws.data_ptr(),
                                              dx.data_ptr(), dwv.data_ptr(), dwh.data_ptr(), dws.data_ptr(), ops._DT[x.dtype], N, C, H, W, K,
                                              wsb.data_ptr(), nbb, _stream(x.device))
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "slak_dwconv2d_tri_backward")
            return dx, dwv, dwh, dws
        dx = None
    if need_dx:
        if tri_dgrad:
            dx = torch.empty_like(x)
            L = _lib.lib()
            with _on(x.device):
                _lib.check(L.slak_dwconv2d_tri_backward_data(dyv.data_ptr(), dyh.data_ptr(), dys.data_ptr(), wv.data_ptr(),
                                                             wh.data_ptr(), ws.data_ptr(), dx.data_ptr(), ops._DT[x.dtype],
                                                             N, C, H, W, K, _stream(x.device)), "slak_dwconv2d_tri_backward_data")
        else:
            dx = ops.dwconv2d_backward_data(dyv, wv)
            if accumulate_dgrad:
                ops.dwconv2d_backward_data_accumulate(dyh, wh, dx)   # autograd's two adds folded into the kernels' copy-out
                ops.dwconv2d_backward_data_accumulate(dys, ws, dx)
            else:
                dx += ops.dwconv2d_backward_data(dyh, wh)
                dx += ops.dwconv2d_backward_data(dys, ws)
    dwv = dwh = dws = None
    if all(need_w) and fused_tri_wgrad and plan[3]:          # one launch for the three weight gradients (x fetched once)
        L = _lib.lib()
        dt = ops._DT[x.dtype]
        dwv, dwh, dws = (torch.empty_like(w, dtype=torch.float32) for w in (wv, wh, ws))
        wsb, nbb = _workspace(plan[3], x.device)
        with _on(x.device):
            rc = L.slak_dwconv2d_tri_backward_filter(dyv.data_ptr(), dyh.data_ptr(), dys.data_ptr(), x.data_ptr(), dwv.data_ptr(),
                                                     dwh.data_ptr(), dws.data_ptr(), dt, N, C, H, W, K, wsb.data_ptr(), nbb, _stream(x.device))
        if rc == _lib.ERR_UNSUPPORTED:
            dwv = dwh = dws = None
        else:
            _lib.check(rc, "slak_dwconv2d_tri_backward_filter")
    if dwv is None and need_w[0] and need_w[2] and fused_tri_wgrad and plan[4]:      # K x 5 and 5 x 5 in one launch (x fetched and shifted once)
        L = _lib.lib()
        dt = ops._DT[x.dtype]
        dwv, dws = (torch.empty_like(w, dtype=torch.float32) for w in (wv, ws))
        wsb, nbb = _workspace(plan[4], x.device)
        with _on(x.device):
            rc = L.slak_dwconv2d_pair_backward_filter(dyv.data_ptr(), dys.data_ptr(), x.data_ptr(), dwv.data_ptr(), dws.data_ptr(),
                                                      dt, N, C, H, W, K, wsb.data_ptr(), nbb, _stream(x.device))
        if rc == _lib.ERR_UNSUPPORTED:
            dwv = dws = None
        else:
            _lib.check(rc, "slak_dwconv2d_pair_backward_filter")
            dwh = ops.dwconv2d_backward_filter(dyh, x, wh) if need_w[1] else None
    if dwv is None:
        dwv = ops.dwconv2d_backward_filter(dyv, x, wv) if need_w[0] else None
        dwh = ops.dwconv2d_backward_filter(dyh, x, wh) if need_w[1] else None
        dws = ops.dwconv2d_backward_filter(dys, x, ws) if need_w[2] else None
    return dx, dwv, dwh, dws


class _TriDwConv(torch.autograd.Function):
    """The three branch convolutions of a decomposed large-kernel block -- LoRA1 (K x 5), LoRA2 (5 x K), small_conv (5 x 5) on the
    same input (models/SLaK.py:82-100) -- as one autograd node.  Where the one-launch kernels exist (slak_dwconv2d_tri_*) the input is
    read once forward and the three input gradients are summed in the accumulator; elsewhere the per-branch kernels run and the
    gradients are added here."""

    @staticmethod
    def forward(ctx, x, wv, wh, ws, want_stats=False):
        yv, yh, ys, stats, ctx.tri_dgrad = _tri_forward_impl(x, wv, wh, ws, want_stats)
        ctx.save_for_backward(x, wv, wh, ws)
        # (the three statistics outputs are non-differentiable; autograd would still hand backward() a ZERO tensor for each of them --
        # three fill launches per block and step, 54 of the 56 FillFunctor launches of a SLaK-T step)
        ctx.set_materialize_grads(False)
        if want_stats:
            # the sums travel as three non-differentiable outputs: one [rows][C][6] array (three-branch launch) seen through three offsets,
            # three [rows_b][C][2] arrays (per-branch launches), or three empty tensors
            stats = _stats_triple(stats, x.device)
            ctx.mark_non_differentiable(*stats)
            return (yv, yh, ys) + tuple(stats)
        return yv, yh, ys

    @staticmethod
    def backward(ctx, dyv, dyh, dys, *_dstats):
        x, wv, wh, ws = ctx.saved_tensors
        dx, dwv, dwh, dws = _tri_backward_impl(x, wv, wh, ws, dyv, dyh, dys, ctx.tri_dgrad, ctx.needs_input_grad[0], ctx.needs_input_grad[1:4])
        return dx, dwv, dwh, dws, None


def _stats_triple(stats, device):
    if stats is None:
        e = torch.empty(0, device=device)
        return (e, e.clone(), e.clone())
    if not isinstance(stats, tuple):
        return (stats, stats[:, :, 2:], stats[:, :, 4:])
    return stats


bn_stats_in_conv = os.environ.get("SLAK_BN_STATS_IN_CONV", "1") != "0"   # three-branch forward launches also leave the branch BatchNorms' batch sums


def tri_dwconv(x, w_vertical, w_horizontal, w_small, want_stats=False):
    """(y_v, y_h, y_s) = depthwise conv of x with the (C,1,K,5), (C,1,5,K) and (C,1,5,5) filters (stride 1, 'same' padding).
    want_stats (1 or 2): a fourth result, a triple of tensors for branch_bn3(..., stats=); with 2 they hold the partial batch sums of the
    stored outputs where the launches gather them (views [rows][C][>=2]: sum, sum of squares in the first two columns), otherwise they are empty."""
    r = _TriDwConv.apply(x, w_vertical, w_horizontal, w_small, want_stats)
    return (r[0], r[1], r[2], tuple(r[3:])) if want_stats else r


def tri_dwconv_sum(x, w_vertical, w_horizontal, w_small, bias=None):
    """y = conv(x, w_v) + conv(x, w_h) + conv(x, w_s) (+ bias): the re-parameterised (BatchNorms folded) decomposed block at inference
    (ReparamLargeKernelConv.merge_kernel).  No autograd.  Where the one-launch kernel exists the three branches are
    summed in its accumulator: a forward correlation with w equals the data-gradient kernel run on the spatially flipped filter,
    so this is slak_dwconv2d_tri_backward_data with x as all three inputs; elsewhere three launches and two adds."""
    from . import ops
    with torch.no_grad():
        _chk(x, "input")
        N, C, H, W = x.shape
        K = w_vertical.shape[2]
        L = _lib.lib()
        dt = ops._DT.get(x.dtype)
        ws = [w.detach().float().contiguous() for w in (w_vertical, w_horizontal, w_small)]
        if (dt is not None and ws[0].shape == (C, 1, K, 5) and ws[1].shape == (C, 1, 5, K) and ws[2].shape == (C, 1, 5, 5)
                and L.slak_dwconv2d_tri_supported_op(dt, N, C, H, W, K, 1) == 1):      # (it IS the data-gradient kernel: same policy as the training node)
            fl = [w.flip(2, 3).contiguous() for w in ws]
            y = torch.empty_like(x)
            with _on(x.device):
                _lib.check(L.slak_dwconv2d_tri_backward_data(x.data_ptr(), x.data_ptr(), x.data_ptr(), fl[0].data_ptr(), fl[1].data_ptr(),
                                                             fl[2].data_ptr(), y.data_ptr(), dt, N, C, H, W, K, _stream(x.device)),
                           "slak_dwconv2d_tri_backward_data")
        else:
            y = ops.dwconv2d_forward(x, ws[0])
            y += ops.dwconv2d_forward(x, ws[1])
            y += ops.dwconv2d_forward(x, ws[2])
        if bias is not None:
            y += bias.detach().to(y.dtype).view(1, -1, 1, 1)
        return y


def ln_nchw_to_nhwc(x, weight, bias, eps=1e-6):
    return _LnNchwToNhwc.apply(x, weight, bias, eps)


def scale_residual(shortcut, z, gamma, sample_scale=None, emit_lowp=False):
    """Returns `out` (fp32 NCHW), or (out, out_bf16) with emit_lowp."""
    return _ScaleResidual.apply(shortcut, z, gamma, sample_scale, emit_lowp)


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
        dw = linear_wgrad(dy2.contiguous(), x2.contiguous()) if dy2.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 else None
        if dw is None and S > 1:
            dw = torch.bmm(dy2.view(S, M // S, -1).transpose(1, 2), x2.view(S, M // S, -1)).sum(0, dtype=torch.float32)
        elif dw is None:
            dw = torch.mm(dy2.t(), x2).float()
        db = dy2.sum(0, dtype=torch.float32) if ctx.has_bias else None
        return dx, dw, db


def linear_splitk(x, weight, bias):
    return _LinearSplitK.apply(x, weight, bias)


# ------------------------------------------------------------------------------------------------------------------
# out = BN1(y1) + BN2(y2) + BN3(y3)   (models/SLaK.py:38-47, :92-95)
def _ptr3(ts):
    import ctypes
    arr = (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
    return arr


class BnCounterPool:
    """The step counters (``num_batches_tracked``) of many BatchNorms as views of ONE int64 tensor that is bumped by a single launch per
    forward pass (nn.BatchNorm / the reference bump one 0-dim tensor per BatchNorm: 54 launches per SLaK-T step; ``torch._foreach_add_``
    takes its slow path for them).  Each BatchNorm keeps its own 0-dim ``num_batches_tracked`` buffer (state-dict layout unchanged); the
    owner (slak_model.SLaK) brackets a training forward with ``begin_forward()`` / ``end_forward()``; inside it the first fused BatchNorm op
    to call ``bump_once()`` bumps every counter, the others find it done.  OUTSIDE such a forward (a stage or a block called on its own, a
    feature extractor, a second sub-forward) ``bump_once()`` returns False and the op bumps just its own three counters the ordinary way.
    A buffer that stopped being a view (``module.to()`` re-creates buffers) is not covered any more and is bumped the ordinary way too."""

    def __init__(self, bns):
        self.bns = list(bns)
        self.flat = torch.stack([bn.num_batches_tracked.detach().reshape(()) for bn in self.bns]).contiguous()
        for i, bn in enumerate(self.bns):
            bn._buffers["num_batches_tracked"] = self.flat[i]
            bn._slak_ctr_pool = self
            bn._slak_ctr_index = i
        self.active = False       # inside a forward that the owner manages
        self.pending = False      # ... whose one bump has not happened yet

    def _is_view(self, bn):
        t = bn._buffers.get("num_batches_tracked")
        return t is not None and t.device == self.flat.device and t.data_ptr() == self.flat.data_ptr() + 8 * bn._slak_ctr_index

    def covers(self, bns):
        return all(getattr(bn, "_slak_ctr_pool", None) is self and self._is_view(bn) for bn in bns)

    def intact(self):
        return self._is_view(self.bns[0]) and self._is_view(self.bns[-1])

    def begin_forward(self):
        self.active = True
        self.pending = True

    def end_forward(self):
        self.active = False
        self.pending = False

    def bump_once(self):
        """True: the counters of this forward are taken care of (by this call or an earlier one); False: not in a managed forward."""
        if not self.active:
            return False
        if self.pending:
            self.flat.add_(1)
            self.pending = False
        return True


_allreduce_sum_opts = {}


def _sync_bn_all_reduce(buf, group, async_op=False):
    """The SyncBatchNorm statistics exchange of one block and direction (6C + 1 doubles forward, 4C floats backward), in place, SUM.
    async_op: returns the work handle right after the issue (the caller launches independent kernels, then waits); else the collective is
    ordered on the current stream (ProcessGroupNCCL runs a synchronous collective ON the caller's stream: no stream hand-off; gloo blocks the host).
    Goes to the process group object directly -- dist.all_reduce's Python-side argument checks and logging wrapper are ~15 us per call, 36 calls per
    SLaK-T step, 72 per SLaK-B step; same collective, same result."""
    import torch.distributed as dist
    try:
        opts = _allreduce_sum_opts.get(async_op)
        if opts is None:
            opts = dist.AllreduceOptions()
            opts.reduceOp = dist.ReduceOp.SUM
            if hasattr(opts, "asyncOp"):
                opts.asyncOp = bool(async_op)
            _allreduce_sum_opts[async_op] = opts
        work = group.allreduce([buf], opts)
    except (AttributeError, TypeError):                              # (a group object without the C++ method: the public entry point)
        return dist.all_reduce(buf, group=group, async_op=async_op)
    if async_op:
        return work
    if work is not None:
        work.wait()
    return None


_sync_bn_all_reduce_impl = _sync_bn_all_reduce                        # (a test that records the exchange's order replaces the module attribute)


def _runner_exchange(group):
    """What the C++ runner gets for the statistics exchange: None (single process), the ProcessGroup itself (the runner issues the all-reduce from
    C++), or -- when a test has patched _sync_bn_all_reduce, or SLAK_PG_DIRECT=0 -- a callable through this module."""
    if group is None:
        return None
    if _sync_bn_all_reduce is _sync_bn_all_reduce_impl and _pg_direct and hasattr(group, "allreduce"):
        return group
    return lambda buf, async_op: _sync_bn_all_reduce(buf, group, async_op=async_op)


def _bn3_forward_impl(y1, y2, y3, gam, bet, bns, group, pre):
    """Training-mode statistics of the three branch BatchNorms (cross-rank when ``group`` is given: one all-reduce of 6C+1 doubles), running-stat
    update, fused scale / shift / add.  -> (out, stats [6C] for the backward, count, count_dev)"""
    for t, n in ((y1, "y1"), (y2, "y2"), (y3, "y3")):
        _chk(t, n, torch.bfloat16)
    N, C, H, W = y1.shape
    P = H * W
    dev = y1.device
    L = _lib.lib()
    rmean = [bn.running_mean for bn in bns]; rvar = [bn.running_var for bn in bns]
    eps = float(bns[0].eps)
    momentum = bns[0].momentum
    ctrs = [bn.num_batches_tracked for bn in bns if bn.track_running_stats and bn.num_batches_tracked is not None]
    pool = getattr(bns[0], "_slak_ctr_pool", None)
    if pool is not None and len(ctrs) == len(bns) and pool.covers(bns) and pool.bump_once():
        pass                                                  # every pooled counter of the model in one launch per managed forward pass
    elif ctrs:
        torch._foreach_add_(ctrs, 1)
    if momentum is None:                                     # cumulative moving average, as nn.BatchNorm
        momentum = 1.0 / float(bns[0].num_batches_tracked.item())
    ws, nb = _workspace(_bn3_ws_bytes(N, C), dev)
    coef = torch.empty(C * 4, dtype=torch.float32, device=dev)
    stats = torch.empty(C * 6, dtype=torch.float32, device=dev)
    out = torch.empty_like(y1)
    if group is None:                                        # single process: sums, finalise and apply without the exchange step (3 launches)
        with _on(dev):
            _lib.check(L.slak_bn3_forward_local(y1.data_ptr(), y2.data_ptr(), y3.data_ptr(), _ptr3(gam), _ptr3(bet), _ptr3(rmean), _ptr3(rvar),
                                                eps, float(momentum), 1 if bns[0].track_running_stats else 0, coef.data_ptr(), stats.data_ptr(),
                                                out.data_ptr(), N, C, P, ws.data_ptr() if ws is not None else None, nb, _stream(dev),
                                                _ptr3(list(pre)) if pre is not None else None,
                                                (ctypes.c_int * 3)(*[int(t.shape[0]) for t in pre]) if pre is not None else None,
                                                int(pre[0].stride(1)) if pre is not None else 0),
                       "slak_bn3_forward_local")
        return out, stats, float(N * P), None
    sums = torch.empty(C * 6 + 1, dtype=torch.float64, device=dev)      # sum y_b, sum y_b^2 per channel as doubles + the element count
    pre_args = ((_ptr3(list(pre)), (ctypes.c_int * 3)(*[int(t.shape[0]) for t in pre]), int(pre[0].stride(1))) if pre is not None
                else (None, None, 0))                                 # the conv launches' rows feed the exchange buffer: no read pass
    with _on(dev):
        _lib.check(L.slak_bn3_forward_sums_counted(y1.data_ptr(), y2.data_ptr(), y3.data_ptr(), sums.data_ptr(), N, C, P,
                                                   ws.data_ptr() if ws is not None else None, nb, _stream(dev), *pre_args), "slak_bn3_forward_sums_counted")
    count = float(N * P)                                     # (element 6C of the buffer: written by the same launch)
    _sync_bn_all_reduce(sums, group)
    count_dev = sums[C * 6:]                                 # global element count, stays on the device (no host sync)
    with _on(dev):
        _lib.check(L.slak_bn3_forward_apply(y1.data_ptr(), y2.data_ptr(), y3.data_ptr(), sums.data_ptr(), count,
                                            count_dev.data_ptr(), _ptr3(gam), _ptr3(bet), _ptr3(rmean), _ptr3(rvar), eps, float(momentum), 1,
                                            1 if bns[0].track_running_stats else 0,
                                            coef.data_ptr(), stats.data_ptr(), out.data_ptr(), N, C, P, _stream(dev)), "slak_bn3_forward_apply")
    return out, stats, count, count_dev


_bn3_ws_cache = {}


def _bn3_ws_bytes(N, C):
    key = (N, C)
    v = _bn3_ws_cache.get(key)
    if v is None:
        v = _bn3_ws_cache[key] = int(_lib.lib().slak_bn3_workspace_bytes(N, C))
    return v


def _bn3_backward_impl(dout, y1, y2, y3, gs, stats, group, count, count_dev, between=None):
    """-> (d1, d2, d3, dgamma [3][C], dbeta [3][C]).  ``between``: work that does not depend on the statistics exchange (the block's pointwise
    weight gradients); with a process group it is launched AFTER the all-reduce of the backward sums has been issued (asynchronously) and
    BEFORE the apply pass waits for it -- the collective's latency hides behind those launches; single process: simply runs first."""
    N, C, H, W = y1.shape
    P = H * W
    dev = y1.device
    dout = dout.contiguous()
    if dout.dtype != torch.bfloat16:
        dout = dout.to(torch.bfloat16)
    L = _lib.lib()
    ws, nb = _workspace(_bn3_ws_bytes(N, C), dev)
    bcoef = torch.empty(C * 9, dtype=torch.float32, device=dev)
    dgamma = torch.empty(3, C, dtype=torch.float32, device=dev)
    dbeta = torch.empty(3, C, dtype=torch.float32, device=dev)
    d1, d2, d3 = torch.empty_like(y1), torch.empty_like(y2), torch.empty_like(y3)
    if group is None:
        if between is not None:
            between()
        with _on(dev):
            _lib.check(L.slak_bn3_backward_local(dout.data_ptr(), y1.data_ptr(), y2.data_ptr(), y3.data_ptr(), stats.data_ptr(), _ptr3(gs),
                                                 bcoef.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), d1.data_ptr(), d2.data_ptr(), d3.data_ptr(),
                                                 N, C, P, ws.data_ptr() if ws is not None else None, nb, _stream(dev)), "slak_bn3_backward_local")
        return d1, d2, d3, dgamma, dbeta
    lsums = torch.empty(C * 4, dtype=torch.float32, device=dev)
    gsums = torch.empty(C * 4, dtype=torch.float32, device=dev)      # the all-reduce's buffer, written by the same launch as the local sums
    with _on(dev):
        _lib.check(L.slak_bn3_backward_sums_dup(dout.data_ptr(), y1.data_ptr(), y2.data_ptr(), y3.data_ptr(), stats.data_ptr(), lsums.data_ptr(),
                                                gsums.data_ptr(), N, C, P, ws.data_ptr() if ws is not None else None, nb, _stream(dev)),
                   "slak_bn3_backward_sums_dup")
    work = _sync_bn_all_reduce(gsums, group, async_op=between is not None and _bn_bwd_async)
    if between is not None:
        between()                                            # launched behind the collective's issue, in front of its wait
        if work is not None:
            work.wait()
    with _on(dev):
        _lib.check(L.slak_bn3_backward_apply(dout.data_ptr(), y1.data_ptr(), y2.data_ptr(), y3.data_ptr(), gsums.data_ptr(),
                                             lsums.data_ptr(), count, count_dev.data_ptr() if count_dev is not None else None,
                                             stats.data_ptr(), _ptr3(gs),
                                             bcoef.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                             d1.data_ptr(), d2.data_ptr(), d3.data_ptr(), N, C, P, _stream(dev)), "slak_bn3_backward_apply")
    return d1, d2, d3, dgamma, dbeta


class _BranchBN3(torch.autograd.Function):
    """Training-mode batch statistics (cross-rank when ``group`` is given: one all-reduce of 6C+1 floats forward and 4C backward,
    instead of SyncBatchNorm's three all_gathers + three all_reduces per block), running-stat update, fused scale/shift/add."""

    @staticmethod
    def forward(ctx, y1, y2, y3, g1, b1, g2, b2, g3, b3, bns, group, pre=None):
        out, stats, ctx.count, ctx.count_dev = _bn3_forward_impl(y1, y2, y3, [g1, g2, g3], [b1, b2, b3], bns, group, pre)
        ctx.save_for_backward(y1, y2, y3, g1, g2, g3, stats)
        ctx.group = group
        return out

    @staticmethod
    def backward(ctx, dout):
        y1, y2, y3, g1, g2, g3, stats = ctx.saved_tensors
        d1, d2, d3, dgamma, dbeta = _bn3_backward_impl(dout, y1, y2, y3, [g1, g2, g3], stats, ctx.group, ctx.count, ctx.count_dev)
        return d1, d2, d3, dgamma[0], dbeta[0], dgamma[1], dbeta[1], dgamma[2], dbeta[2], None, None, None


def _bn3_pre_rows(stats, C):
    """The conv launches' statistics rows in the form slak_bn3_forward_* take them, or None (SyncBatchNorm too: the rows feed the all-reduce buffer)."""
    if (stats is not None and len(stats) == 3
            and all(t.dim() == 3 and t.shape[0] > 0 and t.shape[1] == C and t.dtype == torch.float32 and t.stride(2) == 1
                    and t.stride(0) == t.shape[1] * t.stride(1) for t in stats) and len({t.stride(1) for t in stats}) == 1):
        return tuple(stats)
    return None


def _bn3_group(bn1):
    import torch.distributed as dist
    if isinstance(bn1, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized():
        pg = bn1.process_group if bn1.process_group is not None else dist.group.WORLD
        if dist.get_world_size(pg) > 1 or _force_bn_exchange:         # (bench.py --force-dist: the exchange path on one rank, so that it executes on RCCL)
            return pg
    return None


def branch_bn3(y1, y2, y3, bn1, bn2, bn3, stats=None):
    """``bn1(y1) + bn2(y2) + bn3(y3)`` for three nn.BatchNorm2d / nn.SyncBatchNorm modules (their parameters and buffers are used and
    updated in place).  Training: batch statistics (synchronised across the default/``process_group`` ranks for SyncBatchNorm);
    eval: running statistics."""
    import torch.distributed as dist
    bns = (bn1, bn2, bn3)
    if bn1.training or not bn1.track_running_stats:
        group = _bn3_group(bn1)
        pre = _bn3_pre_rows(stats, y1.shape[1])
        return _BranchBN3.apply(y1, y2, y3, bn1.weight, bn1.bias, bn2.weight, bn2.bias, bn3.weight, bn3.bias, bns, group, pre)
    # eval: one apply pass with coefficients from the running statistics (no autograd needed for the statistics)
    N, C, H, W = y1.shape
    L = _lib.lib()
    coef = torch.empty(C * 4, dtype=torch.float32, device=y1.device)
    out = torch.empty_like(y1)
    with _on(y1.device):
        _lib.check(L.slak_bn3_forward_apply(y1.data_ptr(), y2.data_ptr(), y3.data_ptr(), None, 0.0, None,
                                            _ptr3([b.weight for b in bns]), _ptr3([b.bias for b in bns]),
                                            _ptr3([b.running_mean for b in bns]), _ptr3([b.running_var for b in bns]),
                                            float(bn1.eps), 0.0, 0, 0, coef.data_ptr(), None, out.data_ptr(), N, C, H * W, _stream(y1.device)),
                   "slak_bn3_forward_apply")
    return out


# ------------------------------------------------------------------------------------------------------------------
# channels_first LayerNorm of the stem / downsample layers (models/SLaK.py:256-261)
_SDT = {torch.float32: _lib.SLAK_F32, torch.bfloat16: _lib.SLAK_BF16}


class _LnChannelsFirst(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_dtype):
        _chk(x, "x"); _chk(weight, "weight", torch.float32); _chk(bias, "bias", torch.float32)
        if x.dtype not in _SDT or out_dtype not in _SDT:
            raise TypeError("x / out must be float32 or bfloat16")
        N, C, H, W = x.shape
        y = torch.empty((N, C, H, W), dtype=out_dtype, device=x.device)
        mean = torch.empty((N, H * W), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L = _lib.lib()
        with _on(x.device):
            _lib.check(L.slak_ln_channels_first_forward(x.data_ptr(), _SDT[x.dtype], weight.data_ptr(), bias.data_ptr(), y.data_ptr(), _SDT[out_dtype],
                                                        mean.data_ptr(), rstd.data_ptr(), N, C, H * W, float(eps), _stream(x.device)),
                       "slak_ln_channels_first_forward")
        ctx.save_for_backward(x, weight, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        x, weight, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        g = g.contiguous()
        if g.dtype not in _SDT:
            g = g.float()
        dx = torch.empty_like(x)
        dw = torch.empty_like(weight); db = torch.empty_like(weight)
        L = _lib.lib()
        ws, nb = _workspace(L.slak_ln_cf_workspace_bytes(N, C, H * W), x.device)
        with _on(x.device):
            _lib.check(L.slak_ln_channels_first_backward(g.data_ptr(), _SDT[g.dtype], x.data_ptr(), _SDT[x.dtype], weight.data_ptr(), mean.data_ptr(),
                                                         rstd.data_ptr(), dx.data_ptr(), dw.data_ptr(), db.data_ptr(), N, C, H * W,
                                                         ws.data_ptr() if ws is not None else None, nb, _stream(x.device)), "slak_ln_channels_first_backward")
        return dx, dw, db, None, None


def ln_channels_first(x, weight, bias, eps=1e-6, out_dtype=torch.float32):
    return _LnChannelsFirst.apply(x, weight, bias, eps, out_dtype)


class _LnChannelsFirstPair(torch.autograd.Function):
    """The stem's LayerNorm with two outputs: y (fp32) and bf16(y) -- the copy the first block's depthwise convs read under autocast (what
    `inputs.to(bfloat16)` inside the block would make) -- and a backward that takes both gradients and adds them while loading
    (slak_ln_channels_first_backward_pair): no cast pass in front of the block, no `grad + grad_lowp` pass behind it."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        _chk(x, "x"); _chk(weight, "weight", torch.float32); _chk(bias, "bias", torch.float32)
        N, C, H, W = x.shape
        y = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
        y16 = torch.empty((N, C, H, W), dtype=torch.bfloat16, device=x.device)
        mean = torch.empty((N, H * W), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L = _lib.lib()
        with _on(x.device):
            _lib.check(L.slak_ln_channels_first_forward_pair(x.data_ptr(), _SDT[x.dtype], weight.data_ptr(), bias.data_ptr(), y.data_ptr(), _lib.SLAK_F32,
                                                             y16.data_ptr(), mean.data_ptr(), rstd.data_ptr(), N, C, H * W, float(eps), _stream(x.device)),
                       "slak_ln_channels_first_forward_pair")
        ctx.save_for_backward(x, weight, mean, rstd)
        return y, y16

    @staticmethod
    def backward(ctx, g, g16):
        x, weight, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        if g is None:
            g, g16 = (torch.zeros_like(x, dtype=torch.float32) if g16 is None else g16.float()), None
        g = g.contiguous()
        if g.dtype != torch.float32:
            g = g.float()
        L = _lib.lib()
        if g16 is not None:
            g16 = g16.contiguous()
            if g16.dtype != torch.bfloat16 or not L.slak_ln_channels_first_backward_pair_supported(_lib.SLAK_F32, _SDT[x.dtype], N, C, H * W):
                g, g16 = g + g16, None                                  # shapes the one-pass kernel does not cover: autograd's own addition
        dx = torch.empty_like(x)
        dw = torch.empty_like(weight); db = torch.empty_like(weight)
        ws, nb = _workspace(L.slak_ln_cf_workspace_bytes(N, C, H * W), x.device)
        with _on(x.device):
            _lib.check(L.slak_ln_channels_first_backward_pair(g.data_ptr(), _lib.SLAK_F32, g16.data_ptr() if g16 is not None else None, x.data_ptr(),
                                                              _SDT[x.dtype], weight.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(), dw.data_ptr(),
                                                              db.data_ptr(), N, C, H * W, ws.data_ptr() if ws is not None else None, nb, _stream(x.device)),
                       "slak_ln_channels_first_backward_pair")
        return dx, dw, db, None


def ln_channels_first_pair(x, weight, bias, eps=1e-6):
    """(LN(x) in fp32, its bf16 copy): see _LnChannelsFirstPair."""
    return _LnChannelsFirstPair.apply(x, weight, bias, eps)


# ------------------------------------------------------------------------------------------------------------------
# Downsample layers (models/SLaK.py:285-311): LayerNorm(channels_first) -> Conv2d(kernel_size=2, stride=2) as ONE LayerNorm kernel that
# writes the conv's GEMM operand (the 2x2 patch matrix, bf16) + batched library GEMMs that produce / consume NCHW directly: no MIOpen
# implicit-GEMM launch, no NCHW<->NHWC transposes around it, no fp32 -> bf16 cast of the LayerNorm output.
def ln_patch_covers(x):
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()):
        return False
    N, C, H, W = x.shape
    return bool(_lib.lib().slak_ln_patch_supported(N, C, H, W))


downsample_2d_gemm = os.environ.get("SLAK_DS_2D", "1") != "0"      # A/B switch: the downsample convs' data gradient as one 2-D GEMM


def channel_sums(dy3):
    """sum over n and p of a contiguous bf16 [N, C, P] gradient -> fp32 [C] (the bias gradient of the stem / downsample convolutions),
    slak_channel_sums_bf16: fixed summation order."""
    N, C, P = dy3.shape
    L = _lib.lib()
    out = torch.empty(C, dtype=torch.float32, device=dy3.device)
    nb = int(L.slak_channel_sums_workspace_bytes(C))
    part = torch.empty(nb, dtype=torch.uint8, device=dy3.device)       # own buffer: the partial sums are read by a second launch
    with _on(dy3.device):
        _lib.check(L.slak_channel_sums_bf16(dy3.data_ptr(), out.data_ptr(), N, C, P, part.data_ptr(), nb, _stream(dy3.device)), "slak_channel_sums_bf16")
    return out


class _DownsampleLnConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln_w, ln_b, conv_w, conv_b, eps):
        _chk(x, "x", torch.float32); _chk(ln_w, "ln weight", torch.float32); _chk(ln_b, "ln bias", torch.float32)
        N, C, H, W = x.shape
        Co = conv_w.shape[0]
        P4 = (H // 2) * (W // 2)
        a = torch.empty((N, P4, 4 * C), dtype=torch.bfloat16, device=x.device)
        mean = torch.empty((N, H * W), dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L = _lib.lib()
        with _on(x.device):
            _lib.check(L.slak_ln_patch_forward(x.data_ptr(), ln_w.data_ptr(), ln_b.data_ptr(), a.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                               N, C, H, W, float(eps), _stream(x.device)), "slak_ln_patch_forward")
        wp = conv_w.detach().permute(0, 2, 3, 1).reshape(Co, 4 * C).to(torch.bfloat16)       # Wp[co][(kh*2+kw)*C + c]
        if conv_b is not None:                                                                  # bias inside the GEMM (fp32 accumulate, one rounding):
            y = torch.empty((N, Co, P4), dtype=torch.bfloat16, device=x.device)                 # the accumulator starts at bf16(bias) (one 16-byte-store fill,
            with _on(x.device):                                                                 # not baddbmm's strided copy of the broadcast bias)
                _lib.check(L.slak_fill_channel_bias_bf16(conv_b.detach().float().contiguous().data_ptr(), y.data_ptr(), N, Co, P4, _stream(x.device)),
                           "slak_fill_channel_bias_bf16")
            y.baddbmm_(wp.unsqueeze(0).expand(N, Co, 4 * C), a.transpose(1, 2))
        else:
            y = torch.matmul(wp, a.transpose(1, 2))                                            # [N, Co, P4]: NCHW
        ctx.save_for_backward(x, ln_w, mean, rstd, a, wp)
        ctx.has_bias = conv_b is not None
        return y.view(N, Co, H // 2, W // 2)

    @staticmethod
    def backward(ctx, dy):
        x, ln_w, mean, rstd, a, wp = ctx.saved_tensors
        N, C, H, W = x.shape
        Co = wp.shape[0]
        P4 = (H // 2) * (W // 2)
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy3 = dy.view(N, Co, P4)
        # weight gradient: the reduction runs over n and the pixels -> dY in pixel-major order (one copy), then the row-reduction GEMM kernel
        # (the library's heuristic picks 64x64 macro tiles for K = N*P': 0.24 / 0.11 / 0.05 ms for the three layers against ~0.04 each)
        if Co % 8 == 0:
            dy_t = torch.empty((N * P4, Co), dtype=torch.bfloat16, device=dy.device)
            with _on(dy.device):
                _lib.check(_lib.lib().slak_nchw_to_pixel_major_bf16(dy3.data_ptr(), dy_t.data_ptr(), N, Co, P4, _stream(dy.device)), "slak_nchw_to_pixel_major_bf16")
        else:
            dy_t = dy3.transpose(1, 2).reshape(N * P4, Co)
        # dL/da in the patch layout [N, P4, 4C]: ONE [N*P4, Co] x [Co, 4C] product on the pixel-major copy (the batched form -- 128 products with
        # M = P4 each -- ran at 0.23-0.35 PFLOP/s)
        da = torch.mm(dy_t, wp).view(N, P4, 4 * C) if downsample_2d_gemm else torch.matmul(dy3.transpose(1, 2), wp)
        dwp = linear_wgrad(dy_t, a.view(N * P4, 4 * C))
        if dwp is None:
            dwp = torch.mm(dy3.permute(1, 0, 2).reshape(Co, N * P4), a.view(N * P4, 4 * C)).float()
        dconv_w = dwp.view(Co, 2, 2, C).permute(0, 3, 1, 2).contiguous()
        dconv_b = channel_sums(dy3) if ctx.has_bias else None
        dx = torch.empty_like(x)
        dlw = torch.empty_like(ln_w); dlb = torch.empty_like(ln_w)
        L = _lib.lib()
        ws, nb = _workspace(L.slak_block_tail_workspace_bytes(N, C, H * W), x.device)
        da = da.contiguous()
        with _on(x.device):
            _lib.check(L.slak_ln_patch_backward(da.data_ptr(), x.data_ptr(), ln_w.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dx.data_ptr(),
                                                dlw.data_ptr(), dlb.data_ptr(), N, C, H, W, ws.data_ptr() if ws is not None else None, nb,
                                                _stream(x.device)), "slak_ln_patch_backward")
        return dx, dlw, dlb, dconv_w, dconv_b, None


class _StemConv(torch.autograd.Function):
    """Conv2d(in_chans, C, kernel_size=4, stride=4) of the fp32 input image under bf16 autocast as patch matrix + batched GEMMs (NCHW result)."""

    @staticmethod
    def forward(ctx, x, conv_w, conv_b):
        _chk(x, "x", torch.float32)
        N, Ci, H, W = x.shape
        Co = conv_w.shape[0]
        P16 = (H // 4) * (W // 4)
        a = torch.empty((N, P16, Ci * 16), dtype=torch.bfloat16, device=x.device)
        L = _lib.lib()
        if (L.slak_stem_conv_forward_supported(N, Ci, H, W, Co) and conv_w.dtype == torch.float32 and conv_w.is_contiguous()
                and (conv_b is None or (conv_b.dtype == torch.float32 and conv_b.is_contiguous()))):
            y = torch.empty((N, Co, H // 4, W // 4), dtype=torch.bfloat16, device=x.device)    # patches, product and bias in one pass from the image
            with _on(x.device):
                _lib.check(L.slak_stem_conv_forward(x.data_ptr(), conv_w.data_ptr(), conv_b.data_ptr() if conv_b is not None else None, a.data_ptr(),
                                                    y.data_ptr(), N, Ci, H, W, Co, _stream(x.device)), "slak_stem_conv_forward")
            ctx.save_for_backward(a)
            ctx.shape = (Co, Ci, conv_b is not None)
            return y
        with _on(x.device):
            _lib.check(L.slak_stem_patchify(x.data_ptr(), a.data_ptr(), N, Ci, H, W, _stream(x.device)), "slak_stem_patchify")
        wp = conv_w.detach().reshape(Co, Ci * 16).to(torch.bfloat16)
        if conv_b is not None:
            y = torch.baddbmm(conv_b.detach().to(torch.bfloat16).view(1, Co, 1).expand(N, Co, P16), wp.unsqueeze(0).expand(N, Co, Ci * 16), a.transpose(1, 2))
        else:
            y = torch.matmul(wp, a.transpose(1, 2))
        ctx.save_for_backward(a)
        ctx.shape = (Co, Ci, conv_b is not None)
        return y.view(N, Co, H // 4, W // 4)

    @staticmethod
    def backward(ctx, dy):
        (a,) = ctx.saved_tensors
        Co, Ci, has_bias = ctx.shape
        N, P16, K = a.shape
        dy = dy.contiguous()
        if dy.dtype != torch.bfloat16:
            dy = dy.to(torch.bfloat16)
        dy3 = dy.view(N, Co, P16)
        L = _lib.lib()
        if L.slak_stem_wgrad_supported(N, Co, P16, K):                                   # dw and db in one pass over dy and a, fp32 accumulation
            dw = torch.empty((Co, K), dtype=torch.float32, device=dy.device)
            db = torch.empty(Co, dtype=torch.float32, device=dy.device) if has_bias else None
            ws, nb = _workspace(int(L.slak_stem_wgrad_workspace_bytes(N, Co, P16, K)), dy.device)
            with _on(dy.device):
                _lib.check(L.slak_stem_wgrad(dy3.data_ptr(), a.data_ptr(), dw.data_ptr(), db.data_ptr() if has_bias else None, N, Co, P16, K,
                                             ws.data_ptr(), nb, _stream(dy.device)), "slak_stem_wgrad")
            return None, dw.view(Co, Ci, 4, 4), db
        dw = torch.bmm(dy3, a).sum(0, dtype=torch.float32).view(Co, Ci, 4, 4)          # per-image products (K = P16 each), fp32 sum over the batch
        db = channel_sums(dy3) if has_bias else None
        return None, dw, db


def stem_conv(x, conv_w, conv_b):
    return _StemConv.apply(x, conv_w, conv_b)


def downsample_ln_conv(x, ln_w, ln_b, conv_w, conv_b, eps=1e-6):
    """conv2d(LN_channels_first(x), conv_w, conv_b, stride=2) for a 2x2 kernel, fp32 NCHW x -> bf16 NCHW (what autocast returns)."""
    return _DownsampleLnConv.apply(x, ln_w, ln_b, conv_w, conv_b, eps)


# ------------------------------------------------------------------------------------------------------------------
import os as _os
import weakref as _weakref

# bf16 copies of the fp32 master weights the pointwise GEMMs read, refreshed in ONE multi-tensor launch per optimizer step
# (autocast casts each weight with its own kernel each step: 72 launches per SLaK-T step).  Keyed by parameter identity,
# validated by the tensor version counter (optimizer steps and Masking's kernels bump it).
# Off by default: a weight changed in place through `.data` (which has its own version counter) would go unnoticed; the training
# loops of the reference (optimizer.step() + Masking) are safe, and bench.py turns it on.
_lowp_cache = {}
cache_lowp_weights = False


def lowp_param(p):
    if not cache_lowp_weights:
        return p.to(torch.bfloat16)
    key = id(p)
    e = _lowp_cache.get(key)
    if e is not None and e[0]() is p and e[2].device == p.device:
        if e[1] != p._version:
            _refresh_lowp()
        return e[2]
    with torch.no_grad():
        c = p.detach().to(torch.bfloat16)
    _lowp_cache[key] = [_weakref.ref(p), p._version, c]
    return c


def _refresh_lowp():
    srcs, dsts, dead = [], [], []
    for key, e in _lowp_cache.items():
        p = e[0]()
        if p is None:
            dead.append(key)
        elif e[1] != p._version and e[2].device == p.device:
            srcs.append(p.detach()); dsts.append(e[2]); e[1] = p._version
    for key in dead:
        del _lowp_cache[key]
    if srcs:
        with torch.no_grad():
            torch._foreach_copy_(dsts, srcs)


# Transposed bf16 copies of the pointwise weights (what dz . W2 and dy1 . W1 read), cached by the parameter's version like the copies above and refreshed --
# ALL stale ones in ONE launch (slak_transpose_bf16_batch) -- at the first use after an optimizer step.  Without the cache (the default outside bench.py) a
# call makes its own transposed copy, as autograd's path does.
_lowp_t_cache = {}


def lowp_param_t(p):
    if not cache_lowp_weights or p.dim() != 2 or p.shape[0] % 8 or p.shape[1] % 8 or not p.is_cuda:
        return p.to(torch.bfloat16).t().contiguous()
    src = lowp_param(p)
    e = _lowp_t_cache.get(id(p))
    if e is not None and e[0]() is p and e[2].device == p.device and e[1] == p._version:
        return e[2]
    if e is None or e[0]() is not p or e[2].device != p.device:
        _lowp_t_cache[id(p)] = e = [_weakref.ref(p), None, torch.empty((p.shape[1], p.shape[0]), dtype=torch.bfloat16, device=p.device)]
    # refresh every stale transpose of this device in one launch (their bf16 sources are current: lowp_param() above refreshed them all)
    jobs, dead = [], []
    for key, ee in _lowp_t_cache.items():
        q = ee[0]()
        if q is None:
            dead.append(key)
        elif ee[1] != q._version and ee[2].device == p.device:
            jobs.append((lowp_param(q), ee))
    for key in dead:
        del _lowp_t_cache[key]
    import ctypes
    n = len(jobs)
    srcs = (ctypes.c_void_p * n)(*[sq.data_ptr() for sq, _ in jobs])
    dsts = (ctypes.c_void_p * n)(*[ee[2].data_ptr() for _, ee in jobs])
    rows = (ctypes.c_int * n)(*[sq.shape[0] for sq, _ in jobs])
    cols = (ctypes.c_int * n)(*[sq.shape[1] for sq, _ in jobs])
    with _on(p.device):
        _lib.check(_lib.lib().slak_transpose_bf16_batch(srcs, dsts, rows, cols, n, _stream(p.device)), "slak_transpose_bf16_batch")
    for sq, ee in jobs:
        ee[1] = ee[0]()._version
        ee[2]._slak_stamp = ee[1]                                    # (w1_fragments keys its packed copy on it)
    return e[2]


def w1_fragments(w1t):
    """W1^T [96][384] bf16 -> the fragment-major copy slak_linear_nt_gelu_bwd_dt reads ([pair][k-step][row tile][lane half][row][8 k]: 1 KB per
    load, every byte used); kept on the cached transpose it was made from until lowp_param_t refreshes that one."""
    stamp = getattr(w1t, "_slak_stamp", None)
    c = getattr(w1t, "_slak_frag", None)
    if stamp is not None and c is not None and c[0] == stamp:
        return c[1]
    if not (w1t.is_cuda and w1t.dtype == torch.bfloat16 and w1t.is_contiguous() and tuple(w1t.shape) == (96, 384)):
        raise _lib.SlakHipError("w1_fragments: W1^T must be a contiguous (96, 384) bfloat16 HIP tensor")
    w1p = torch.empty_like(w1t)                                       # the layout contract lives in the library (slak_pack_w1t_fragments)
    with _on(w1t.device):
        _lib.check(_lib.lib().slak_pack_w1t_fragments(w1t.data_ptr(), w1p.data_ptr(), 384, 96, _stream(w1t.device)), "slak_pack_w1t_fragments")
    if stamp is not None:
        w1t._slak_frag = (stamp, w1p)
    return w1p


_SPLITK_ROWS = int(_os.environ.get("SLAK_SPLITK_ROWS", "6272"))     # rows per split of the weight-gradient GEMMs (0: one plain GEMM)


use_skinny_linear = os.environ.get("SLAK_SKINNY_LINEAR", "1") != "0"      # the streaming kernels of csrc/linear_skinny.hip for the stage-1 pointwise convs (-0.16 ms per SLaK-T step same-box vs the TUNED library GEMMs); 0: library GEMMs everywhere


use_linear_wgrad = os.environ.get("SLAK_LINEAR_WGRAD", "1") != "0"        # csrc/linear_wgrad.hip for the pointwise weight gradients; 0: the library's split-K batched GEMM


def linear_wgrad(dy, x):
    """dW (fp32, [N1][N2]) = dy^T x for bf16 dy [M][N1], x [M][N2] through slak_linear_wgrad; None when the shape is not covered."""
    if not (use_linear_wgrad and dy.is_cuda and dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.is_contiguous() and x.is_contiguous()):
        return None
    M, N1 = dy.shape
    N2 = x.shape[1]
    L = _lib.lib()
    if x.shape[0] != M or not L.slak_linear_wgrad_supported(M, N1, N2):
        return None
    d = torch.empty((N1, N2), dtype=torch.float32, device=dy.device)
    ws, nb = _workspace(L.slak_linear_wgrad_workspace_bytes(M, N1, N2), dy.device)
    with _on(dy.device):
        _lib.check(L.slak_linear_wgrad(dy.data_ptr(), x.data_ptr(), d.data_ptr(), M, N1, N2, ws.data_ptr() if ws is not None else None, nb,
                                       _stream(dy.device)), "slak_linear_wgrad")
    return d


def linear_nt_covers(x, N, gelu=False):
    """Whether slak_linear_nt takes x (..., K) bf16 against a weight with N output features."""
    if not (use_skinny_linear and x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous()):
        return False
    K = x.shape[-1]
    return bool(_lib.lib().slak_linear_nt_supported(x.numel() // K, N, K, 1 if gelu else 0))


def linear_nt(x, wt, bias=None, gelu=False):
    """y = x @ wt.T (+ bias) for bf16 x (..., K), wt (N, K), bias (N,) through slak_linear_nt (the streaming kernels for the skinny
    shapes of the large maps); with gelu=True returns (y, gelu(y)).  None when the shape is not covered (the caller runs the library GEMM)."""
    if not (use_skinny_linear and x.is_cuda and x.dtype == torch.bfloat16 and wt.dtype == torch.bfloat16 and x.is_contiguous() and wt.is_contiguous()):
        return None
    K = x.shape[-1]
    N = wt.shape[0]
    M = x.numel() // K
    L = _lib.lib()
    if wt.shape[1] != K or not L.slak_linear_nt_supported(M, N, K, 1 if gelu else 0):
        return None
    if bias is not None and (bias.dtype != torch.bfloat16 or not bias.is_contiguous()):
        return None
    y = torch.empty(x.shape[:-1] + (N,), dtype=torch.bfloat16, device=x.device)
    g = torch.empty_like(y) if gelu else None
    with _on(x.device):
        _lib.check(L.slak_linear_nt(x.data_ptr(), wt.data_ptr(), bias.data_ptr() if bias is not None else None, y.data_ptr(),
                                    g.data_ptr() if gelu else None, M, N, K, _stream(x.device)), "slak_linear_nt")
    return (y, g) if gelu else y


use_linear_gemm = os.environ.get("SLAK_LINEAR_GEMM", "1") != "0"          # csrc/linear_gemm.hip: pwconv1 + GELU resp. dz W2 + GELU' + bias gradient as ONE launch on stages 2-3 (round 5); 0: library GEMM + elementwise pass


def linear_gemm_gelu(t, w1b, b1b):
    """(y1, a) = (t @ w1b.T + b1b, GELU(y1)) in one launch (slak_linear_gemm, SLAK_EPI_GELU), or None when the shape is not covered."""
    if not (use_linear_gemm and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and w1b.dtype == torch.bfloat16 and w1b.is_contiguous()):
        return None
    K, N = w1b.shape[1], w1b.shape[0]
    M = t.numel() // K
    L = _lib.lib()
    if b1b is None or b1b.dtype != torch.bfloat16 or not L.slak_linear_gemm_supported(M, N, K, 1):
        return None
    y1 = torch.empty(t.shape[:-1] + (N,), dtype=torch.bfloat16, device=t.device)
    a = torch.empty_like(y1)
    with _on(t.device):
        _lib.check(L.slak_linear_gemm(t.data_ptr(), w1b.data_ptr(), b1b.data_ptr(), y1.data_ptr(), a.data_ptr(), None, None, M, N, K, 1, None, 0,
                                      _stream(t.device)), "slak_linear_gemm")
    return y1, a


def linear_gemm_dgelu(dz2, w2t, y12):
    """(dy1, db1) = (round(dz2 @ w2t.T) * gelu'(y12), column sums of dy1) in one launch (SLAK_EPI_DGELU), or None when the shape is not covered."""
    if not (use_linear_gemm and dz2.is_cuda and dz2.dtype == torch.bfloat16 and dz2.is_contiguous() and w2t.dtype == torch.bfloat16 and w2t.is_contiguous()
            and y12.dtype == torch.bfloat16 and y12.is_contiguous()):
        return None
    M, K = dz2.shape
    N = w2t.shape[0]
    L = _lib.lib()
    if tuple(y12.shape) != (M, N) or not L.slak_linear_gemm_supported(M, N, K, 2):
        return None
    dy1 = torch.empty((M, N), dtype=torch.bfloat16, device=dz2.device)
    db1 = torch.empty(N, dtype=torch.float32, device=dz2.device)
    ws, nb = _workspace(int(L.slak_linear_gemm_workspace_bytes(M, N, K, 2)), dz2.device)
    with _on(dz2.device):
        _lib.check(L.slak_linear_gemm(dz2.data_ptr(), w2t.data_ptr(), None, dy1.data_ptr(), None, y12.data_ptr(), db1.data_ptr(), M, N, K, 2,
                                      ws.data_ptr() if ws is not None else None, nb, _stream(dz2.device)), "slak_linear_gemm")
    return dy1, db1


def _mlp_fwd(t, w1, b1, w2, b2):
    F = torch.nn.functional
    w1b, w2b = lowp_param(w1), lowp_param(w2)
    L = _lib.lib()
    C, C4 = w1b.shape[1], w1b.shape[0]
    M = t.numel() // C
    if use_skinny_linear and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and w1b.dtype == torch.bfloat16 and w1b.is_contiguous() and \
            w2b.is_contiguous() and b1 is not None and b2 is not None and L.slak_linear_mlp_fwd_supported(M, C, C4):
        # stage 1: pwconv1, GELU and pwconv2 in ONE pass (W2 lives in the waves' registers; a is not read back)
        y1 = torch.empty(t.shape[:-1] + (C4,), dtype=torch.bfloat16, device=t.device)
        a = torch.empty_like(y1)
        z = torch.empty(t.shape[:-1] + (C,), dtype=torch.bfloat16, device=t.device)
        bb1, bb2 = lowp_param(b1), lowp_param(b2)
        with _on(t.device):
            _lib.check(L.slak_linear_mlp_fwd(t.data_ptr(), w1b.data_ptr(), bb1.data_ptr(), w2b.data_ptr(), bb2.data_ptr(), y1.data_ptr(), a.data_ptr(),
                                             z.data_ptr(), M, C, C4, _stream(t.device)), "slak_linear_mlp_fwd")
        return z, (t, w1b, y1, a, w2b)
    r = linear_nt(t, w1b, lowp_param(b1), gelu=True)            # pwconv1 + GELU in one streaming pass on the large maps
    if r is None and b1 is not None:
        r = linear_gemm_gelu(t, w1b, lowp_param(b1))            # stages 2-3: the GEMM with bias, rounding and GELU in its epilogue (y1 AND a in one launch)
    if r is not None:
        y1, a = r
    else:
        y1 = F.linear(t, w1b, lowp_param(b1))
        a = F.gelu(y1)
    z = linear_nt(a, w2b, lowp_param(b2))
    if z is None:
        z = F.linear(a, w2b, lowp_param(b2))
    return z, (t, w1b, y1, a, w2b)


def _mlp_wgrad(dy, x):
    """dW = dy^T x (fp32) for the pointwise Linear layers: slak_linear_wgrad where it covers the shape, else the library's split-K batched GEMM."""
    d = linear_wgrad(dy, x)
    if d is not None:
        return d
    M = dy.shape[0]
    S = max(1, M // _SPLITK_ROWS) if _SPLITK_ROWS > 0 else 1
    while S > 1 and M % S:
        S -= 1
    if S > 1:
        return torch.bmm(dy.view(S, M // S, -1).transpose(1, 2), x.view(S, M // S, -1)).sum(0, dtype=torch.float32)
    return torch.mm(dy.t(), x).float()


_gelu_ws_cache = {}


def _mlp_bwd_data(saved, dz, wts=None):
    """The data path of the MLP's backward: dz -> dact -> (GELU') dy1 (+ pwconv1's bias gradient) -> dt.  -> (dt, dy1 [M][4C], db1)
    wts: (w1b^T, w2b^T) when the caller holds the cached transposed copies (lowp_param_t), else they are made here."""
    t, w1b, y1, a, w2b = saved
    w1t_of = (lambda: wts[0]) if wts is not None else (lambda: w1b.t().contiguous())
    w2t_of = (lambda: wts[1]) if wts is not None else (lambda: w2b.t().contiguous())
    dz2 = dz.reshape(-1, dz.shape[-1])
    y12 = y1.reshape(-1, y1.shape[-1])
    M = dz2.shape[0]
    dact = None
    L = _lib.lib()
    N4 = w2b.shape[1]
    if use_skinny_linear and dz2.is_contiguous() and y12.is_contiguous() and dz2.dtype == torch.bfloat16 and w2b.dtype == torch.bfloat16 and \
            L.slak_linear_nt_gelu_bwd_supported(M, N4, dz2.shape[1]):
        # stage 1: dz @ W2, GELU' and pwconv1's bias gradient in ONE pass (dact never reaches HBM)
        dy1 = torch.empty((M, N4), dtype=torch.bfloat16, device=dz2.device)
        db1 = torch.empty(N4, dtype=torch.float32, device=dz2.device)
        w2t = w2t_of()
        ws, nb = _workspace(int(L.slak_linear_nt_gelu_bwd_workspace_bytes(M, N4, dz2.shape[1])), dz2.device)
        if L.slak_linear_nt_gelu_bwd_dt_supported(M, N4, dz2.shape[1]):
            # ... and dt = dy1 @ W1 from the dy1 tiles while they are on chip (no second launch, dy1 is not read back)
            w1p = w1_fragments(w1t_of())
            dt = torch.empty((M, dz2.shape[1]), dtype=torch.bfloat16, device=dz2.device)
            with _on(dz2.device):
                _lib.check(L.slak_linear_nt_gelu_bwd_dt(dz2.data_ptr(), w2t.data_ptr(), y12.data_ptr(), w1p.data_ptr(), dy1.data_ptr(), dt.data_ptr(), db1.data_ptr(),
                                                        M, N4, dz2.shape[1], ws.data_ptr(), nb, _stream(dz2.device)), "slak_linear_nt_gelu_bwd_dt")
            return dt.view_as(t), dy1, db1
        with _on(dz2.device):
            _lib.check(L.slak_linear_nt_gelu_bwd(dz2.data_ptr(), w2t.data_ptr(), y12.data_ptr(), dy1.data_ptr(), db1.data_ptr(), M, N4, dz2.shape[1],
                                                 ws.data_ptr(), nb, _stream(dz2.device)), "slak_linear_nt_gelu_bwd")
        dt = linear_nt(dy1, w1t_of()) if linear_nt_covers(dy1, w1b.shape[1]) else None
        dt = (dt if dt is not None else torch.mm(dy1, w1b)).view_as(t)
        return dt, dy1, db1
    r = linear_gemm_dgelu(dz2, w2t_of(), y12) if (dz2.is_contiguous() and y12.is_contiguous()) else None
    if r is not None:                                            # stages 2-3: dz @ W2 with GELU' and pwconv1's bias gradient in the GEMM's epilogue
        dy1, db1 = r
        dt = linear_nt(dy1, w1t_of()) if linear_nt_covers(dy1, w1b.shape[1]) else None
        dt = (dt if dt is not None else torch.mm(dy1, w1b)).view_as(t)
        return dt, dy1, db1
    if linear_nt_covers(dz2, w2b.shape[1]):
        dact = linear_nt(dz2, w2t_of())              # dz @ W2: NT against the (small) transposed weight
    if dact is None:
        dact = torch.mm(dz2, w2b)
    dy1 = torch.empty_like(dact)
    db1 = torch.empty(dact.shape[1], dtype=torch.float32, device=dact.device)
    key = (M, dact.shape[1])
    nbw = _gelu_ws_cache.get(key)
    if nbw is None:
        nbw = _gelu_ws_cache[key] = int(L.slak_gelu_bwd_workspace_bytes(M, dact.shape[1]))
    ws, nb = _workspace(nbw, dact.device)
    with _on(dact.device):
        _lib.check(L.slak_gelu_backward_bias(dact.data_ptr(), y12.data_ptr(), dy1.data_ptr(), db1.data_ptr(), M, dact.shape[1],
                                             ws.data_ptr() if ws is not None else None, nb, _stream(dact.device)), "slak_gelu_backward_bias")
    dt = linear_nt(dy1, w1t_of()) if linear_nt_covers(dy1, w1b.shape[1]) else None
    dt = (dt if dt is not None else torch.mm(dy1, w1b)).view_as(t)
    return dt, dy1, db1


def _mlp_bwd_weights(saved, dz, dy1):
    """The two pointwise weight gradients (they feed nothing else of the block's backward) -> (dw1, dw2)"""
    t, w1b, y1, a, w2b = saved
    dz2 = dz.reshape(-1, dz.shape[-1])
    dw2 = _mlp_wgrad(dz2, a.reshape(-1, a.shape[-1]))
    dw1 = _mlp_wgrad(dy1, t.reshape(-1, t.shape[-1]))
    return dw1, dw2


def _mlp_bwd(saved, dz, db2=None):
    """db2: the column sums of dz when the caller already has them (scale_residual's backward produces them for free)."""
    dt, dy1, db1 = _mlp_bwd_data(saved, dz)
    dw1, dw2 = _mlp_bwd_weights(saved, dz, dy1)
    if db2 is None:
        db2 = dz.reshape(-1, dz.shape[-1]).sum(0, dtype=torch.float32)
    return dt, dw1, db1, dw2, db2


class _MlpSplitK(torch.autograd.Function):
    """pwconv2(gelu(pwconv1(t))) under bf16 autocast (models/SLaK.py:158-160) with (i) both weight gradients as split-K batched
    library GEMMs (see _LinearSplitK) and (ii) the GELU backward fused with pwconv1's bias gradient in one HIP kernel.
    The GEMMs themselves stay hipBLASLt calls."""

    @staticmethod
    def forward(ctx, t, w1, b1, w2, b2):
        z, saved = _mlp_fwd(t, w1, b1, w2, b2)
        ctx.save_for_backward(*saved)
        return z

    @staticmethod
    def backward(ctx, dz):
        return _mlp_bwd(ctx.saved_tensors, dz)


class _MlpScaleResidual(torch.autograd.Function):
    """The whole tail of a block behind the LayerNorm as ONE autograd node: z = pwconv2(gelu(pwconv1(t))), out = shortcut +
    sample_scale * gamma * z (NHWC -> NCHW), optionally with the bf16 copy of `out` for the next block's convs
    (models/SLaK.py:158-165).  One node because the residual kernel's backward produces, besides dz, the column sums of dz --
    pwconv2's bias gradient: they go from one half of this backward to the other as a plain local variable (they used to travel as a
    Python attribute on the dz tensor, which any hook that re-creates the tensor would have dropped)."""

    @staticmethod
    def forward(ctx, shortcut, t, w1, b1, w2, b2, gamma, sample_scale, emit_lowp):
        z, saved = _mlp_fwd(t, w1, b1, w2, b2)
        if z.dtype != torch.bfloat16:
            z = z.to(torch.bfloat16)
        z = z.contiguous()
        out, out16 = _scale_residual_fwd(shortcut, z, gamma, sample_scale, emit_lowp)
        ctx.save_for_backward(*saved, z, gamma, sample_scale)
        ctx.shortcut_dtype = shortcut.dtype
        return (out, out16) if emit_lowp else out

    @staticmethod
    def backward(ctx, dout, dout16=None):
        *saved, z, gamma, sample_scale = ctx.saved_tensors
        dshortcut, dz, dgamma, dzc = _scale_residual_bwd(z, gamma, sample_scale, ctx.shortcut_dtype, dout, dout16)
        dt, dw1, db1, dw2, db2 = _mlp_bwd(saved, dz, db2=dzc)
        return dshortcut, dt, dw1, db1, dw2, db2, dgamma, None, None


def mlp_scale_residual(shortcut, t, w1, b1, w2, b2, gamma, sample_scale=None, emit_lowp=False):
    """out (and, with emit_lowp, its bf16 copy) = shortcut + sample_scale * gamma * pwconv2(gelu(pwconv1(t))) -- see _MlpScaleResidual."""
    return _MlpScaleResidual.apply(shortcut, t, w1, b1, w2, b2, gamma, sample_scale, emit_lowp)


def mlp_splitk(t, w1, b1, w2, b2):
    return _MlpSplitK.apply(t, w1, b1, w2, b2)


# ------------------------------------------------------------------------------------------------------------------
# One autograd node per block (models/SLaK.py:153-166): dw convs x 3 -> BN x 3 + add -> permute + LayerNorm -> pwconv1 -> GELU ->
# pwconv2 -> gamma * + permute + residual.  The same launches as the four nodes above in the same order forward; backward the node owns the ORDER:
# the chain that the SyncBatchNorm statistics exchange waits for (residual kernel -> dz W2 -> GELU' -> dy1 W1 -> LayerNorm backward -> BatchNorm
# sums) is issued first, the all-reduce goes out asynchronously, the two pointwise weight-gradient launches (which nothing in the block waits for)
# run behind it, then the apply pass waits -- the collective's latency hides behind ~80 us of launches instead of stalling the stream.
_runner_mod = False           # False: not looked for yet; None: not there / stale / failed to import
_runner_trace = None          # tests: a callable(str) that the C++ runner calls at the points whose ORDER a test asserts (tests/test_distributed_gpu.py)
_bn_bwd_async = os.environ.get("SLAK_BN_BWD_ASYNC", "0") == "1"            # 1: the backward exchange on the collective's own stream, overlapped with the two pointwise weight gradients (round 4-5 default).
# Round 6 default: on the compute stream like the forward one -- measured on one MI355X over RCCL (--force-dist): 15.77 vs 16.03 ms per SLaK-T step, 25.62 vs 25.94 SLaK-B,
# and ~1.4 ms less host time per step (no Work objects, no waits); the two stream hand-offs per block cost about what the overlap hides
_pg_direct = os.environ.get("SLAK_PG_DIRECT", "0") == "1"                  # 1: the runner calls ProcessGroup::allreduce itself instead of calling back into _sync_bn_all_reduce.  Opt-in: measured on one MI355X over
# RCCL it changes neither the step (16.59 vs 16.46 ms SLaK-T, 25.97 vs 26.11 SLaK-B) nor, beyond the noise, the host time -- dist.all_reduce's Python side is no longer the cost after round 6's direct group call
_force_bn_exchange = os.environ.get("SLAK_FORCE_BN_EXCHANGE", "0") == "1"   # bench.py --force-dist: run the SyncBatchNorm exchange (the all-reduces) at world size 1 too


def _runner_switches():
    """The development switches that select paths only the Python sequence knows: with any of them off their default the runner steps aside, so an
    A/B run measures what its switch says.  Read on every call: a test may flip a switch after the first block has run."""
    return (use_skinny_linear and use_linear_wgrad and use_linear_gemm and _SPLITK_ROWS == 6272 and fused_tri_backward and fused_tri_wgrad
            and bn_stats_in_conv and accumulate_dgrad and os.environ.get("SLAK_BLOCK_RUNNER", "1") != "0")


def _runner():
    """slak_amd/pybind/block_runner.cpp, when `__graft_entry__.build()` / `python -m slak_amd.build --pybind` has built it (never built here: an
    import must not start a compiler): the block's call sequence issued from C++ -- two host calls per block and step.  SLAK_BLOCK_RUNNER=0 keeps
    the Python sequence (same launches, same results).  A module OLDER than libslak_hip.so / include/slak_hip.h / its own source is not loaded
    (it would call raw-pointer entry points with whatever argument lists it was compiled against) -- and one that does load checks
    slak_version() against the SLAK_ABI_VERSION it was compiled with."""
    global _runner_mod
    if _runner_mod is False:
        _runner_mod = None
        try:
            import importlib
            import sys
            import warnings
            from . import build
            path = build.runner_path()
            if os.path.exists(path):
                deps = [build.LIB, os.path.join(build.HERE, "..", "include", "slak_hip.h"), os.path.join(build.HERE, "pybind", "block_runner.cpp")]
                if build._stale(path, [d for d in deps if os.path.exists(d)]):
                    warnings.warn("slak_amd: %s is older than libslak_hip.so / slak_hip.h / block_runner.cpp and is NOT loaded (the Python call sequence "
                                  "runs instead); rebuild with `python -m slak_amd.build --pybind`" % os.path.basename(path))
                else:
                    d = os.path.dirname(path)
                    if d not in sys.path:
                        sys.path.insert(0, d)
                    _runner_mod = importlib.import_module(build.RUNNER_NAME)
        except Exception as e:                                        # (no compiled module for this interpreter / torch, or an ABI mismatch: the Python sequence runs)
            import warnings
            warnings.warn("slak_amd: block runner not loaded (%s: %s)" % (type(e).__name__, e))
            _runner_mod = None
    return _runner_mod if (_runner_mod is not None and _runner_switches()) else None


# ------------------------------------------------------------------------------------------------------------------
# Parameter gradients written where DistributedDataParallel wants them (round 6; main.py:374-376 wraps the model, engine.py:80-88 steps it).
# DDP(gradient_as_bucket_view=True) makes every .grad a view of an all-reduce bucket -- but a step that drops the gradients
# (zero_grad(set_to_none=True), engine.py:86 / torch's default) gets fresh tensors from autograd, and the reducer copies each into its bucket view as
# the parameter becomes ready: one launch per parameter, 312 per SLaK-T step, 1.4 ms of GPU time and the reducer's host time around them
# (profiles/r05_step_breakdown_forcedist.txt).  The bucket views outlive the step, so: `adopt_grad_slots` remembers a parameter's .grad tensor
# once a backward has left a bucket view there (MaskedAdamW.step calls it), and the next backward of the block hands those tensors to the
# C++ runner as the DESTINATIONS of its weight-gradient / reduction launches; the runner returns new tensor objects on the same storage, autograd
# installs them as .grad without a copy, the reducer finds .grad aliasing its view and launches nothing.  Self-healing: a destination that is no
# longer the reducer's view (DDP rebuilds its buckets once after the first iteration) is just a gradient tensor again -- the reducer copies it,
# re-points .grad, and the next adoption picks the new view up.  A parameter whose .grad is still set at backward time (gradient accumulation
# under no_sync(), engine.py:61-66 with update_freq > 1) is never given a destination: autograd adds in place there.
grad_slots_enabled = False    # bench.py / Masking.add_module turn it on for a DDP wrapper with gradient_as_bucket_view=True
_grad_slots = {}              # id(parameter) -> (weakref(parameter), destination tensor)
grad_slot_hits = 0            # destinations handed to the runner so far (tests / bench.py's config line)


_adopted_key = None


def adopt_grad_slots(params, key=None):
    """Remember every parameter's current .grad tensor as the destination of its next gradient (see above).  `key`: anything that changes
    when a .grad pointer does (MaskedAdamW passes its tuple of gradient addresses): the walk is skipped while it repeats.  Entries of dead
    parameters are dropped."""
    import weakref
    global _adopted_key
    if key is not None:
        if key == _adopted_key and _grad_slots:
            return
        _adopted_key = key
    for p in params:
        g = p.grad
        e = _grad_slots.get(id(p))
        if e is not None and e[0]() is not p:
            e = None
            del _grad_slots[id(p)]
        if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape:
            continue
        if e is None or e[1].data_ptr() != g.data_ptr():
            # (the entry dies with its parameter: the destination keeps a whole all-reduce bucket alive)
            _grad_slots[id(p)] = (weakref.ref(p, lambda _r, k=id(p): _grad_slots.pop(k, None)), g)


def drop_grad_slots():
    global _adopted_key
    _grad_slots.clear()
    _adopted_key = None


def enable_grad_slots_for(module):
    """Turn the destinations on when `module` is a DistributedDataParallel wrapper whose gradients are bucket views (a no-op -- and False --
    for anything else: with main.py:374-376's constructor defaults the reducer copies out of .grad whatever it points to)."""
    global grad_slots_enabled
    ddp = getattr(torch.nn.parallel, "DistributedDataParallel", None)
    if ddp is not None and isinstance(module, ddp) and bool(getattr(module, "gradient_as_bucket_view", False)) \
            and os.environ.get("SLAK_GRAD_SLOTS", "1") != "0":
        grad_slots_enabled = True
        return True
    return False


def _grad_destinations(params):
    """The runner's `grad_dst` list for a block's sixteen parameters: the adopted destination where there is one and .grad is unset."""
    global grad_slot_hits
    if not grad_slots_enabled or not _grad_slots:
        return []
    out = []
    for p in params:
        e = _grad_slots.get(id(p))
        out.append(e[1] if (e is not None and p.grad is None and e[0]() is p) else None)
    grad_slot_hits += sum(o is not None for o in out)
    return out


class _BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, x_lowp, wv, wh, ws, g1, b1, g2, b2, g3, b3, lnw, lnb, w1, bb1, w2, bb2, gamma, sample_scale, cfg):
        bns, eps, emit = cfg["bns"], cfg["eps"], cfg["emit_lowp"]
        ctx.runner = False
        R = _runner()
        if (R is not None and bns[0].momentum is not None and x.is_contiguous()
                and all(bn.track_running_stats and bn.num_batches_tracked is not None for bn in bns)):
            w1b, w2b = lowp_param(w1), lowp_param(w2)
            group = _bn3_group(bns[0])                                # SyncBatchNorm: the runner calls back for the two statistics exchanges
            exchange = _runner_exchange(group)
            res = R.block_forward(x, x_lowp, wv, wh, ws, [g1, g2, g3], [b1, b2, b3], [bn.running_mean for bn in bns], [bn.running_var for bn in bns],
                                  float(bns[0].eps), float(bns[0].momentum), True, lnw, lnb, float(eps), w1b, lowp_param(bb1), w2b, lowp_param(bb2),
                                  gamma, sample_scale, bool(emit), exchange)
            if res:                                                   # (empty: the shape has no one-launch path -- the Python sequence knows the fallbacks)
                out, out16, x16, yv, yh, ys, bnstats, s, t, mean, rstd, y1m, a, z, count_dev = res
                pool = getattr(bns[0], "_slak_ctr_pool", None)
                if not (pool is not None and pool.covers(bns) and pool.bump_once()):
                    torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
                wts = (lowp_param_t(w1), lowp_param_t(w2)) if cache_lowp_weights else (None, None)
                ctx.save_for_backward(x16, wv, wh, ws, yv, yh, ys, g1, g2, g3, bnstats, s, lnw, mean, rstd, t, w1b, y1m, a, w2b, z, gamma, sample_scale, *wts)
                ctx.misc = (True, group, float(x.shape[0] * x.shape[2] * x.shape[3]), count_dev, x.dtype, x_lowp is not None)
                ctx.runner = True
                ctx.params = (wv, wh, ws, g1, b1, g2, b2, g3, b3, lnw, lnb, w1, bb1, w2, bb2, gamma)
                ctx.set_materialize_grads(False)
                return (out, out16) if emit else out
        x16 = x_lowp if x_lowp is not None else x.to(torch.bfloat16)
        x16 = x16.contiguous()
        yv, yh, ys, st, tri_dgrad = _tri_forward_impl(x16, wv, wh, ws, 2)
        group = _bn3_group(bns[0])
        pre = _bn3_pre_rows(_stats_triple(st, x.device), x16.shape[1]) if st is not None else None
        s, bnstats, count, count_dev = _bn3_forward_impl(yv, yh, ys, [g1, g2, g3], [b1, b2, b3], bns, group, pre)
        t, mean, rstd = _ln_fwd_impl(s, lnw, lnb, eps)
        z, saved = _mlp_fwd(t, w1, bb1, w2, bb2)
        if z.dtype != torch.bfloat16:
            z = z.to(torch.bfloat16)
        z = z.contiguous()
        out, out16 = _scale_residual_fwd(x, z, gamma, sample_scale, emit)
        wts = (lowp_param_t(w1), lowp_param_t(w2)) if cache_lowp_weights else (None, None)
        ctx.save_for_backward(x16, wv, wh, ws, yv, yh, ys, g1, g2, g3, bnstats, s, lnw, mean, rstd, *saved, z, gamma, sample_scale, *wts)
        ctx.misc = (tri_dgrad, group, count, count_dev, x.dtype, x_lowp is not None)
        ctx.set_materialize_grads(False)
        return (out, out16) if emit else out

    @staticmethod
    def backward(ctx, dout, dout16=None):
        (x16, wv, wh, ws, yv, yh, ys, g1, g2, g3, bnstats, s, lnw, mean, rstd, t, w1b, y1m, a, w2b, z, gamma, sample_scale, w1t, w2t) = ctx.saved_tensors
        tri_dgrad, group, count, count_dev, xdtype, had_lowp = ctx.misc
        if ctx.runner:
            exchange = _runner_exchange(group)
            w1p = None                                                # stage 1: W1^T in fragment order for the launch that also produces dt
            if tuple(w1b.shape) == (384, 96) and w1b.dtype == torch.bfloat16 and _lib.lib().slak_linear_nt_gelu_bwd_dt_supported(t.numel() // 96, 384, 96):
                w1p = w1_fragments(w1t if w1t is not None else w1b.t().contiguous())
            res = _runner_mod.block_backward(
                x16, wv, wh, ws, yv, yh, ys, [g1, g2, g3], bnstats, s, lnw, mean, rstd, t, w1b, y1m, a, w2b, z, gamma, sample_scale, dout, dout16,
                xdtype == torch.bfloat16, had_lowp, count_dev, exchange, _runner_trace, w1t, w2t, w1p, _grad_destinations(ctx.params))
            return (*res, None, None)                                 # dx, dx_lowp, then the sixteen parameter gradients in forward()'s order
        saved = (t, w1b, y1m, a, w2b)
        dshortcut, dz, dgamma, dzc = _scale_residual_bwd(z, gamma, sample_scale, xdtype, dout, dout16)
        dt, dy1, db1 = _mlp_bwd_data(saved, dz, (w1t, w2t) if w1t is not None else None)
        ds, dlnw, dlnb = _ln_bwd_impl(dt, s, lnw, mean, rstd)
        wg = {}

        def weights():
            wg["dw1"], wg["dw2"] = _mlp_bwd_weights(saved, dz, dy1)
        d1, d2, d3, dgam, dbet = _bn3_backward_impl(ds, yv, yh, ys, [g1, g2, g3], bnstats, group, count, count_dev, between=weights)
        need = ctx.needs_input_grad
        dx16, dwv, dwh, dws = _tri_backward_impl(x16, wv, wh, ws, d1, d2, d3, tri_dgrad, need[0] or need[1], need[2:5])
        dx = dxl = None
        if had_lowp:
            dx, dxl = dshortcut, dx16
        elif dx16 is not None:
            dx = dshortcut + dx16                                     # (not in place: dshortcut may be autograd's own incoming tensor)
        else:
            dx = dshortcut
        return (dx, dxl, dwv, dwh, dws, dgam[0], dbet[0], dgam[1], dbet[1], dgam[2], dbet[2], dlnw, dlnb, wg["dw1"], db1, wg["dw2"], dzc, dgamma,
                None, None)


def fused_block(x, x_lowp, wv, wh, ws, bns, lnw, lnb, eps, w1, bb1, w2, bb2, gamma, sample_scale=None, emit_lowp=False):
    """One block of the decomposed SLaK trunk in training mode as ONE autograd node -- see _BlockFn.  ``bns``: the three branch BatchNorm
    modules (their parameters are the differentiable inputs, their buffers are updated in place).  Returns out, or (out, out_bf16)."""
    cfg = {"bns": bns, "eps": eps, "emit_lowp": emit_lowp}
    return _BlockFn.apply(x, x_lowp, wv, wh, ws, bns[0].weight, bns[0].bias, bns[1].weight, bns[1].bias, bns[2].weight, bns[2].bias,
                          lnw, lnb, w1, bb1, w2, bb2, gamma, sample_scale, cfg)
