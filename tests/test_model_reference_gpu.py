"""SURVEY 8(a) row a7 pinned against the REFERENCE model: tests/golden/model_reference.npz holds a narrow SLaK built and run (fp64,
CPU) by the unmodified /root/reference/models/SLaK.py (tests/golden/make_golden.py --only model: timm and the CUDA extension are
shimmed at the import boundary, the extension by the nn.Conv2d the reference's own smoke test compares it with).  The mirror
(slak_amd/slak_model.py) loads that state dict and must reproduce the training-mode forward, every parameter gradient, the
BatchNorm running statistics and the eval-mode forward THROUGH THE HIP PATH -- fp32 with the reference's module composition,
and bf16 autocast with every fused op the bench turns on."""
import ast

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _build(g, gpu, lowp):
    import slak_amd.slak_model as M
    cfg = ast.literal_eval(str(g["cfg"]))
    cfg.pop("res")
    M.use_sync_bn = False
    m = M.SLaK(lowp_dwconv=lowp, **cfg)
    keys = [k[len("state0/"):] for k in g if k.startswith("state0/")]
    m.load_state_dict({k: torch.from_numpy(g["state0/" + k]) for k in keys}, strict=True)
    return m.to(gpu)


def _run(m, g, gpu, autocast):
    x = torch.from_numpy(g["x"]).to(gpu)
    dlogits = torch.from_numpy(g["dlogits"]).to(gpu)
    m.train()
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        logits = m(x)
    (logits.float() * dlogits).sum().backward()
    grads = {n: p.grad.detach().double().cpu().numpy() for n, p in m.named_parameters()}
    running = {k: v.detach().double().cpu().numpy() for k, v in m.state_dict().items() if "running_" in k}
    m.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        ev = m(x)
    return logits.detach().double().cpu().numpy(), grads, running, ev.double().cpu().numpy()


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def _set_fused(on, one_node=False):
    import slak_amd.slak_model as M
    from slak_amd import block_ops
    M.Block.fused_block = bool(on and one_node)
    M.Block.fused_tail = on
    M.ReparamLargeKernelConv.fused_bn = on
    M.ReparamLargeKernelConv.fused_tri = on
    M.LayerNorm.fused_cf = on
    block_ops.cache_lowp_weights = False


def test_fp32_mirror_reproduces_the_reference_model(gpu):
    g = load_golden("model_reference")
    _set_fused(False)
    m = _build(g, gpu, lowp=False)
    logits, grads, running, ev = _run(m, g, gpu, autocast=False)
    assert _rel(logits, g["logits_train"]) <= 2e-4
    assert _rel(ev, g["logits_eval"]) <= 2e-4
    for k, v in running.items():
        assert _rel(v, g["state1/" + k]) <= 1e-4, k
    worst = max((_rel(v, g["grad/" + n]), n) for n, v in grads.items())
    assert worst[0] <= 2e-3, worst


def test_fp32_mirror_on_the_matrix_cores_stays_within_the_reference_tolerance(gpu):
    """The same fp32 run with the opt-in two-term split (slak_set_fp32_matrix_cores): every dw conv of the narrow model's maps runs three bf16
    MFMAs per product.  End to end the logits stay within 1e-3 of the reference model's fp64 run (north star; measured ~1e-4), gradients 5e-3."""
    from slak_amd import _lib, ops
    g = load_golden("model_reference")
    _set_fused(False)
    m = _build(g, gpu, lowp=False)
    prev = ops.allow_fp32_matrix_cores(True)
    try:
        logits, grads, running, ev = _run(m, g, gpu, autocast=False)
        assert b"f32 split" in _lib.lib().slak_debug_last_kernel()
    finally:
        ops.allow_fp32_matrix_cores(prev)
    print("fp32 split: logits rel err %.2e (train) %.2e (eval)" % (_rel(logits, g["logits_train"]), _rel(ev, g["logits_eval"])))
    assert _rel(logits, g["logits_train"]) <= 1e-3 and _rel(ev, g["logits_eval"]) <= 1e-3
    worst = max((_rel(v, g["grad/" + n]), n) for n, v in grads.items())
    assert worst[0] <= 5e-3, worst


@pytest.mark.parametrize("fused", [False, True])
def test_bf16_mirror_stays_within_the_bf16_tolerance_of_the_reference_model(fused, gpu):
    g = load_golden("model_reference")
    _set_fused(fused)
    try:
        m = _build(g, gpu, lowp=True)
        logits, grads, running, ev = _run(m, g, gpu, autocast=True)
    finally:
        _set_fused(False)
    # north star: 1e-2 per bf16 op.  This is a chain of 5 blocks (15 dw convs, 10 GEMMs, 17 normalisations) compared END TO END with the reference
    # model's fp64 run, and it still meets the per-op figure on the logits (measured on MI355X: 0.0037-0.0043 train, 0.0040-0.0061 eval; running
    # statistics 0.009; worst parameter gradient 0.022, median 0.006) -- the bounds below are those with a factor ~2 for other seeds of the silicon
    assert _rel(logits, g["logits_train"]) <= 1e-2
    assert _rel(ev, g["logits_eval"]) <= 1e-2
    for k, v in running.items():
        assert _rel(v, g["state1/" + k]) <= 1.5e-2, k
    errs = sorted(((_rel(v, g["grad/" + n]), n) for n, v in grads.items()), reverse=True)
    assert errs[0][0] <= 4e-2, errs[:5]
    assert np.median([e for e, _ in errs]) <= 1e-2


def test_one_autograd_node_per_block_reproduces_the_four_node_block_bit_for_bit(gpu):
    """block_ops.fused_block: the block's four fused ops as ONE autograd node issue the same launches on the same operands --
    logits, every parameter gradient and the BatchNorm running statistics must be IDENTICAL to the four-node composition, not merely close."""
    g = load_golden("model_reference")
    res = []
    for one_node in (False, True):
        _set_fused(True, one_node)
        try:
            torch.manual_seed(5)                                      # (drop path draws)
            m = _build(g, gpu, lowp=True)
            for st in m.stages:                                       # the hand-off of the bf16 copy between blocks, as bench.py runs it
                for blk in list(st)[:-1]:
                    blk.emit_lowp = True
            res.append(_run(m, g, gpu, autocast=True))
        finally:
            _set_fused(False)
    (l0, g0, r0, e0), (l1, g1, r1, e1) = res
    assert np.array_equal(l0, l1) and np.array_equal(e0, e1)
    assert set(g0) == set(g1)
    bad = [n for n in g0 if not np.array_equal(g0[n], g1[n])]
    assert not bad, bad[:5]
    assert all(np.array_equal(r0[k], r1[k]) for k in r0)


def test_block_runner_issues_the_same_launches_as_the_python_node(gpu):
    """slak_amd/pybind/block_runner.cpp (the block's call sequence in C++, built by __graft_entry__.build()): bit-identical logits, gradients and
    running statistics with and without it, on the narrow reference model AND on a stage-1-sized block (where the streaming pointwise kernels and
    the one-launch weight gradient of the 56 x 56 class run)."""
    from slak_amd import block_ops
    import slak_amd.slak_model as M
    if block_ops._runner() is None:
        pytest.skip("block runner module not built")
    g = load_golden("model_reference")
    res = []
    for use in (False, True):
        _set_fused(True, True)
        saved = block_ops._runner_mod
        if not use:
            block_ops._runner_mod = None
        try:
            torch.manual_seed(5)
            m = _build(g, gpu, lowp=True)
            for st in m.stages:
                for blk in list(st)[:-1]:
                    blk.emit_lowp = True
            res.append(_run(m, g, gpu, autocast=True))
            torch.manual_seed(6)
            blk = M.Block(96, drop_path=0.0, layer_scale_init_value=0.5, kernel_size=(51, 5), Decom=True, bn=True, lowp_dwconv=True).to(gpu)
            blk.train()
            x = torch.randn(4, 96, 56, 56, device=gpu, requires_grad=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(x)
            y.float().pow(2).mean().backward()
            res[-1] = res[-1] + (y.detach().double().cpu().numpy(), x.grad.double().cpu().numpy(),
                                 {n: p.grad.double().cpu().numpy() for n, p in blk.named_parameters()})
        finally:
            block_ops._runner_mod = saved
            _set_fused(False)
    (l0, g0, r0, e0, y0, dx0, bg0), (l1, g1, r1, e1, y1, dx1, bg1) = res
    assert np.array_equal(l0, l1) and np.array_equal(e0, e1) and np.array_equal(y0, y1) and np.array_equal(dx0, dx1)
    bad = [n for n in g0 if not np.array_equal(g0[n], g1[n])] + [n for n in bg0 if not np.array_equal(bg0[n], bg1[n])]
    assert not bad, bad[:5]
    assert all(np.array_equal(r0[k], r1[k]) for k in r0)
