"""Row n2 (north_star: "the module surface ... stays identical so the new op drops into models/SLaK.py"): the REFERENCE's own
Python modules, unmodified, executing on libslak_hip.so on the GPU.

  * depthwise_conv2d_implicit_gemm.py (the op module, :8 imports the pybind extension) on the compiled
    ``_depthwise_conv2d_implicit_gemm_C`` of slak_amd/pybind/frontend_hip.cpp;
  * models/SLaK.py (:17 imports the op module; get_conv2d :21-22, ReparamLargeKernelConv :60-100, Block, SLaK) on (a) that reference op
    module and (b) slak_amd/depthwise_conv2d_implicit_gemm.py -- against tests/golden/model_reference.npz, which the same file wrote
    in fp64 on the CPU with nn.Conv2d in the op's place (tests/golden/make_golden.py --only model).

The modules come from /root/reference when it exists and from the bytecode ``oracle/ref_modules.build()`` leaves under
oracle/_ref/ otherwise (the GPU box): no skip on either side."""
import ast
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

LIBDIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "slak_amd", "lib")


def _ref_modules():
    from oracle import ref_modules
    for n in ref_modules.SOURCES:
        assert ref_modules.available(n), "run __graft_entry__.build() where /root/reference exists: oracle/_ref/%s.pyc missing" % n
    return ref_modules


def _pybind_ready():
    from slak_amd import build
    if not os.path.exists(build.pybind_path()):
        build.build_pybind()
    return os.path.dirname(build.pybind_path())


@pytest.mark.parametrize("force_bytecode", [False, True])
def test_reference_bytecode_is_built_and_loads(force_bytecode, monkeypatch):
    """CPU: the two modules resolve (source, or the oracle/_ref bytecode the GPU box uses); the op module binds the compiled extension."""
    rm = _ref_modules()
    if force_bytecode:
        rm.build()
        monkeypatch.setattr(rm, "FORCE_BYTECODE", True)
    op = rm.load_dwconv_module(_pybind_ready())
    assert op.__spec__.origin.endswith(".pyc") == (force_bytecode or not os.path.isdir(rm.REF))
    assert op.__all__ == ["DepthWiseConv2dImplicitGEMM"]
    m = op.DepthWiseConv2dImplicitGEMM(6, (51, 5), bias=False)
    assert isinstance(m, torch.nn.Conv2d) and tuple(m.weight.shape) == (6, 1, 51, 5) and m.padding == (0, 0)
    ref = rm.load_slak_model(op)
    ref.use_sync_bn = False
    net = ref.SLaK(in_chans=3, num_classes=5, depths=[1, 1, 1, 1], dims=[4, 8, 8, 8], kernel_size=[13, 11, 9, 7, 5], Decom=True, bn=True)
    convs = [mod for mod in net.modules() if isinstance(mod, op.DepthWiseConv2dImplicitGEMM)]
    assert len(convs) == 12                                   # 4 blocks x (LoRA1, LoRA2, small_conv): get_conv2d built the op (models/SLaK.py:21-22)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.float16, 1e-2)])
@pytest.mark.parametrize("C,H,W,k", [(6, 28, 28, (51, 5)), (5, 14, 14, (5, 47)), (4, 56, 56, (5, 51)), (3, 9, 11, (7, 7)), (4, 7, 7, (13, 5))])
def test_reference_op_module_runs_on_the_hip_library(C, H, W, k, dtype, tol, gpu):
    """depthwise_conv2d_implicit_gemm.py:52-66 (module) and :14-49 (autograd functions), fwd + bwd, vs the C oracle and vs nn.Conv2d
    with the same state dict -- the reference's own smoke test (:69-82) on every branch shape of a SLaK block."""
    import oracle
    op = _ref_modules().load_dwconv_module(_pybind_ready())
    torch.manual_seed(42)
    m = op.DepthWiseConv2dImplicitGEMM(C, k, bias=True).to(gpu)
    if dtype != torch.float32:
        m = m.to(dtype)
    x = torch.randn(3, C, H, W, device=gpu).to(dtype).requires_grad_(True)
    dy = torch.randn(3, C, H, W, device=gpu).to(dtype)
    y = m(x)
    y.backward(dy)
    assert y.dtype == dtype and x.grad.dtype == dtype
    xn, dyn = x.detach().float().cpu().numpy(), dy.float().cpu().numpy()
    wn, bn = m.weight.detach().float().cpu().numpy(), m.bias.detach().float().cpu().numpy()
    want_y = oracle.dwconv2d_fwd(xn, wn) + bn[None, :, None, None]
    want_dx = oracle.dwconv2d_bwd_data(dyn, wn)
    want_dw = oracle.dwconv2d_bwd_filter(dyn, xn, *k)
    for got, ref, what in ((y, want_y, "y"), (x.grad, want_dx, "dx"), (m.weight.grad, want_dw, "dw")):
        err = np.abs(got.detach().double().cpu().numpy() - ref).max()
        assert err <= tol * max(1.0, np.abs(ref).max()), (what, err)
    # the reference's own check: an nn.Conv2d with the same state dict (CPU, fp64 here)
    m2 = torch.nn.Conv2d(C, C, k, padding=(k[0] // 2, k[1] // 2), groups=C).double()
    m2.load_state_dict({n: v.detach().double().cpu() for n, v in m.state_dict().items()})
    y2 = m2(x.detach().double().cpu())
    assert (y.detach().double().cpu() - y2).abs().max().item() <= tol * max(1.0, y2.abs().max().item())


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["reference_op_module", "slak_amd_op_module"])
def test_reference_slak_model_runs_on_the_hip_library(which, gpu):
    """models/SLaK.py, unmodified, fp32, every DepthWiseConv2dImplicitGEMM a HIP launch: training-mode forward, every parameter
    gradient, BatchNorm running statistics and the eval-mode forward against the fixture the same file produced in fp64 on the CPU."""
    rm = _ref_modules()
    if which == "reference_op_module":
        op = rm.load_dwconv_module(_pybind_ready())
    else:
        import slak_amd.depthwise_conv2d_implicit_gemm as op
    ref = rm.load_slak_model(op)
    ref.use_sync_bn = False                                   # single process (models/SLaK.py:19, :24-28)
    g = load_golden("model_reference")
    cfg = ast.literal_eval(str(g["cfg"]))
    cfg.pop("res")
    model = ref.SLaK(**cfg)
    n_ops = sum(isinstance(mod, op.DepthWiseConv2dImplicitGEMM) for mod in model.modules())
    assert n_ops == 3 * sum(cfg["depths"])
    keys = [k[len("state0/"):] for k in g if k.startswith("state0/")]
    model.load_state_dict({k: torch.from_numpy(g["state0/" + k]) for k in keys}, strict=True)
    model = model.to(gpu)
    x = torch.from_numpy(g["x"]).to(gpu)
    dlogits = torch.from_numpy(g["dlogits"]).to(gpu)
    calls = {"n": 0}
    if which == "slak_amd_op_module":
        from slak_amd import ops
        real = ops.dwconv2d_forward

        def counting(*a, **k):
            calls["n"] += 1
            return real(*a, **k)
        ops.dwconv2d_forward = counting
    try:
        model.train()
        logits = model(x)
        (logits * dlogits).sum().backward()
    finally:
        if which == "slak_amd_op_module":
            ops.dwconv2d_forward = real
            assert calls["n"] == n_ops                        # every conv of the reference model went through the C ABI
    assert _rel(logits.detach().double().cpu().numpy(), g["logits_train"]) <= 2e-4
    worst = max((_rel(p.grad.detach().double().cpu().numpy(), g["grad/" + n]), n) for n, p in model.named_parameters())
    assert worst[0] <= 2e-3, worst
    for k, v in model.state_dict().items():
        if "running_" in k:
            assert _rel(v.detach().double().cpu().numpy(), g["state1/" + k]) <= 1e-4, k
    model.eval()
    with torch.no_grad():
        ev = model(x)
    assert _rel(ev.double().cpu().numpy(), g["logits_eval"]) <= 2e-4
