"""Tensor-level wrappers over the C ABI: the functions the reference's pybind module exports
(cutlass/examples/19_large_depthwise_conv2d_torch_extension/frontend.cpp:3-16), plus bf16.

torch is plumbing here: it owns device memory and the current HIP stream; the arithmetic is in
slak_amd/csrc/*.hip.  Outputs are allocated through the torch caching allocator like the reference
does (``torch::empty_like``: forward_fp32.cu:206).
"""
import torch

from . import _lib

_DT = {torch.float32: _lib.SLAK_F32, torch.float16: _lib.SLAK_F16, torch.bfloat16: _lib.SLAK_BF16}


def _dt(t, name):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError("Only support fp32, fp16 and bf16, get {} for {}".format(t.dtype, name))


def _check_tensor(t, name):
    # same conditions the reference enforces with TORCH_CHECK (forward_fp32.cu:194-196, :203-204)
    if not t.is_cuda:
        raise RuntimeError("%s must be a CUDA/HIP tensor" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)


def _dims(x, w):
    if x.dim() != 4 or w.dim() != 4 or w.shape[1] != 1 or w.shape[0] != x.shape[1]:
        raise RuntimeError("expected x (N,C,H,W) and depthwise weight (C,1,kh,kw), got %s and %s" % (tuple(x.shape), tuple(w.shape)))
    N, C, H, W = x.shape
    return N, C, H, W, w.shape[2], w.shape[3]


_ws_cache = {}


def _workspace(nbytes, device):
    """Per-(device, stream) scratch, grown on demand; stream-ordered reuse is safe because every kernel
    that touches it is enqueued on that same stream."""
    if nbytes == 0:
        return None, 0
    key = (device.index, _stream(device))
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf, buf.numel()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream(device):
    """The raw handle of torch's current stream on `device` (what every C-ABI entry point takes).  The two private torch._C calls
    skip building a torch.cuda.Stream object per launch (~5 us each, ~100 launches per step); without them: the public API."""
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL = _NullCtx()


def _on(device):
    """`with _on(device):` == `with torch.cuda.device(device):`, without the device switch (and its two runtime calls) when `device`
    is already current -- the one-process-per-GPU case."""
    if _cur_device is not None and device.index is not None and _cur_device() == device.index:
        return _NULL
    return torch.cuda.device(device)


def allow_fp32_matrix_cores(allow=True):
    """fp32 tensors through the bf16 matrix cores (two-term split, slak_set_fp32_matrix_cores): off by default like
    torch.backends.cudnn.allow_tf32 gates TF32 -- the fp32 path is then the exact VALU kernels.  Returns the previous PROCESS-WIDE setting (never a per-thread
    override that happens to be active: `prev = allow_fp32_matrix_cores(x); ...; allow_fp32_matrix_cores(prev)` is safe anywhere)."""
    L = _lib.lib()
    prev = bool(L.slak_get_fp32_matrix_cores())
    _lib.check(L.slak_set_fp32_matrix_cores(1 if allow else 0), "slak_set_fp32_matrix_cores")
    return prev


class fp32_matrix_cores:
    """``with fp32_matrix_cores(True):`` -- the calls of THIS thread inside the block run fp32 tensors on the matrix cores (two-term split)
    whatever the process-wide switch says; ``False`` forces the exact kernels; nests."""

    def __init__(self, enabled=True):
        self.mode = 1 if enabled else 0

    def __enter__(self):
        import ctypes
        prev = ctypes.c_int(-1)
        _lib.check(_lib.lib().slak_set_fp32_matrix_cores_thread(self.mode, ctypes.byref(prev)), "slak_set_fp32_matrix_cores_thread")
        self.prev = prev.value
        return self

    def __exit__(self, *exc):
        _lib.check(_lib.lib().slak_set_fp32_matrix_cores_thread(self.prev, None), "slak_set_fp32_matrix_cores_thread")
        return False


def dwconv2d_forward(x, w, out_dtype=None):
    _check_tensor(x, "input"); _check_tensor(w, "weight")
    N, C, H, W, kh, kw = _dims(x, w)
    y = torch.empty_like(x, dtype=out_dtype or x.dtype)
    L = _lib.lib()
    ws, nb = _workspace(L.slak_dwconv2d_workspace_bytes(_lib.OP_FWD, N, C, H, W, kh, kw, _dt(x, "input")), x.device)
    with _on(x.device):
        _lib.check(L.slak_dwconv2d_forward(x.data_ptr(), _dt(x, "input"), w.data_ptr(), _dt(w, "weight"), y.data_ptr(), _dt(y, "output"),
                                           N, C, H, W, kh, kw, ws.data_ptr() if ws is not None else None, nb, _stream(x.device)),
                   "slak_dwconv2d_forward")
    return y


def dwconv2d_forward_stats(x, w):
    """(y, stats): dwconv2d_forward that also returns the partial batch sums [rows][C][2] (sum y, sum y^2 of the stored outputs) for the
    BatchNorm behind the conv; stats is None where the launch does not gather them (y is then the plain forward)."""
    import ctypes
    _check_tensor(x, "input"); _check_tensor(w, "weight")
    N, C, H, W, kh, kw = _dims(x, w)
    L = _lib.lib()
    if x.dtype == torch.bfloat16 and w.dtype == torch.float32:
        y = torch.empty_like(x)
        stats = torch.empty((4 * N, C, 2), dtype=torch.float32, device=x.device)
        rows = ctypes.c_int(0)
        with _on(x.device):
            rc = L.slak_dwconv2d_forward_stats(x.data_ptr(), _dt(x, "input"), w.data_ptr(), _dt(w, "weight"), y.data_ptr(), _dt(y, "output"),
                                               stats.data_ptr(), 4 * N, ctypes.byref(rows), N, C, H, W, kh, kw, _stream(x.device))
        if rc == _lib.OK:
            return y, stats[:rows.value]
        if rc != _lib.ERR_UNSUPPORTED:
            _lib.check(rc, "slak_dwconv2d_forward_stats")
    return dwconv2d_forward(x, w), None


def dwconv2d_backward_data(dy, w, out_dtype=None):
    _check_tensor(dy, "grad"); _check_tensor(w, "weight")
    N, C, H, W, kh, kw = _dims(dy, w)
    dx = torch.empty_like(dy, dtype=out_dtype or dy.dtype)
    L = _lib.lib()
    ws, nb = _workspace(L.slak_dwconv2d_workspace_bytes(_lib.OP_BWD_DATA, N, C, H, W, kh, kw, _dt(dy, "grad")), dy.device)
    with _on(dy.device):
        _lib.check(L.slak_dwconv2d_backward_data(dy.data_ptr(), _dt(dy, "grad"), w.data_ptr(), _dt(w, "weight"), dx.data_ptr(), _dt(dx, "dx"),
                                                 N, C, H, W, kh, kw, ws.data_ptr() if ws is not None else None, nb, _stream(dy.device)),
                   "slak_dwconv2d_backward_data")
    return dx


def dwconv2d_backward_data_accumulate(dy, w, dx):
    """dx += data gradient, in place (slak_dwconv2d_backward_data_accumulate); where the kernel for this shape cannot accumulate the
    gradient goes to a temporary and is added with a tensor add (what autograd does)."""
    _check_tensor(dy, "grad"); _check_tensor(w, "weight"); _check_tensor(dx, "dx")
    N, C, H, W, kh, kw = _dims(dy, w)
    if dx.shape != dy.shape or dx.dtype != dy.dtype:
        raise RuntimeError("dx and grad must have the same shape and dtype")
    L = _lib.lib()
    ws, nb = _workspace(L.slak_dwconv2d_workspace_bytes(_lib.OP_BWD_DATA, N, C, H, W, kh, kw, _dt(dy, "grad")), dy.device)
    with _on(dy.device):
        rc = L.slak_dwconv2d_backward_data_accumulate(dy.data_ptr(), _dt(dy, "grad"), w.data_ptr(), _dt(w, "weight"), dx.data_ptr(), _dt(dx, "dx"),
                                                      N, C, H, W, kh, kw, ws.data_ptr() if ws is not None else None, nb, _stream(dy.device))
    if rc == _lib.ERR_UNSUPPORTED:
        dx += dwconv2d_backward_data(dy, w)
        return dx
    _lib.check(rc, "slak_dwconv2d_backward_data_accumulate")
    return dx


def dwconv2d_backward_filter(dy, x, w):
    """dw, always float32 (backward_filter_fp16.cu:187)."""
    _check_tensor(dy, "grad"); _check_tensor(x, "input"); _check_tensor(w, "weight")
    N, C, H, W, kh, kw = _dims(x, w)
    if dy.shape != x.shape or dy.dtype != x.dtype:
        raise RuntimeError("grad and input must have the same shape and dtype")
    dw = torch.empty((C, 1, kh, kw), dtype=torch.float32, device=x.device)
    L = _lib.lib()
    ws, nb = _workspace(L.slak_dwconv2d_workspace_bytes(_lib.OP_BWD_FILTER, N, C, H, W, kh, kw, _dt(x, "input")), x.device)
    with _on(x.device):
        _lib.check(L.slak_dwconv2d_backward_filter(dy.data_ptr(), _dt(dy, "grad"), x.data_ptr(), _dt(x, "input"), dw.data_ptr(),
                                                   N, C, H, W, kh, kw, ws.data_ptr() if ws is not None else None, nb, _stream(x.device)),
                   "slak_dwconv2d_backward_filter")
    return dw


# ---- the reference pybind module's names (frontend.cpp:3-16), + bf16 ----------------------------
def forward_fp32(x, w): return dwconv2d_forward(x, w)
def backward_data_fp32(dy, w): return dwconv2d_backward_data(dy, w)
def backward_filter_fp32(dy, x, w): return dwconv2d_backward_filter(dy, x, w)
def forward_fp16(x, w): return dwconv2d_forward(x, w)
def backward_data_fp16(dy, w): return dwconv2d_backward_data(dy, w)
def backward_filter_fp16(dy, x, w): return dwconv2d_backward_filter(dy, x, w)
def forward_bf16(x, w): return dwconv2d_forward(x, w)
def backward_data_bf16(dy, w): return dwconv2d_backward_data(dy, w)
def backward_filter_bf16(dy, x, w): return dwconv2d_backward_filter(dy, x, w)
