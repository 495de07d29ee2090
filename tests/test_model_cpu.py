"""The caller mirror (slak_amd/slak_model.py) keeps the reference model's parameter set and names
(models/SLaK.py:186-215; SURVEY.md Appendix A counts)."""
import numpy as np
import torch

import slak_amd.slak_model as M


def test_slak_tiny_parameter_set():
    M.use_sync_bn = False
    m = M.SLaK_tiny(kernel_size=[51, 49, 47, 13, 5], Decom=True)
    assert sum(p.numel() for p in m.parameters()) == 30816232
    maskable = [(n, tuple(p.shape)) for n, p in m.named_parameters() if p.dim() in (2, 4)]
    assert len(maskable) == 95 and sum(int(np.prod(s)) for _, s in maskable) == 30717984
    assert [s for _, s in maskable] == M.slak_mask_set_shapes("tiny")
    lora = [s for n, s in maskable if "large_kernel.LoRA" in n]
    assert len(lora) == 36 and sum(int(np.prod(s)) for s in lora) == 2352960 and lora == M.slak_mask_set_shapes("tiny", only_L=True)
    sd = m.state_dict()
    for k in ("stages.0.0.large_kernel.LoRA1.conv.weight", "stages.0.0.large_kernel.LoRA2.conv.weight",
              "stages.0.0.large_kernel.small_conv.conv.weight", "stages.0.0.large_kernel.LoRA1.bn.running_mean",
              "stages.2.8.pwconv1.weight", "stages.3.2.gamma", "downsample_layers.0.0.weight", "norm.weight", "head.bias"):
        assert k in sd, k
    assert sd["stages.0.0.large_kernel.LoRA1.conv.weight"].shape == (96, 1, 51, 5)
    assert sd["stages.1.0.large_kernel.LoRA2.conv.weight"].shape == (192, 1, 5, 49)
    assert sd["stages.3.0.large_kernel.LoRA1.conv.weight"].shape == (768, 1, 13, 5)
    # trunc_normal_(std=.02) init of every conv / linear, zero bias (models/SLaK.py:217-224)
    w = sd["stages.2.0.pwconv1.weight"]
    assert abs(w.std().item() - 0.02) < 2e-3 and w.abs().max().item() <= 2.0
    assert float(sd["head.bias"].abs().max()) == 0.0


def test_slak_base_dims_and_sync_bn_switch():
    assert len(M.slak_mask_set_shapes("base")) == 185
    only_l = M.slak_mask_set_shapes("base", only_L=True)
    assert len(only_l) == 72 and sum(int(np.prod(s)) for s in only_l) == 7468800        # 2*5*(3*128*51 + 3*256*49 + 27*512*47 + 3*1024*13)
    M.use_sync_bn = True
    blk = M.Block(8, kernel_size=(13, 5), Decom=True)
    assert isinstance(blk.large_kernel.LoRA1.bn, torch.nn.SyncBatchNorm)
    M.use_sync_bn = False
    blk = M.Block(8, kernel_size=(13, 5), Decom=True)
    assert isinstance(blk.large_kernel.LoRA1.bn, torch.nn.BatchNorm2d)
    assert not hasattr(M.Block(8, kernel_size=(5, 5), Decom=True).large_kernel, "small_conv")   # small < kernel only


def test_mirror_has_the_reference_models_parameter_set():
    """tests/golden/model_reference.npz was written by the REFERENCE's models/SLaK.py (make_golden.py --only model): the mirror
    built with the same arguments has the same parameters, in the same order, with the same shapes, and the same buffers."""
    import ast
    from conftest import load_golden
    g = load_golden("model_reference")
    cfg = ast.literal_eval(str(g["cfg"]))
    cfg.pop("res")
    M.use_sync_bn = False
    m = M.SLaK(**cfg)
    names = [str(n) for n in g["names"]]
    assert [n for n, _ in m.named_parameters()] == names
    for n, p in m.named_parameters():
        assert tuple(p.shape) == g["state0/" + n].shape == g["grad/" + n].shape, n
    ref_keys = sorted(k[len("state0/"):] for k in g if k.startswith("state0/"))
    assert sorted(m.state_dict().keys()) == ref_keys
    m.load_state_dict({k: torch.from_numpy(g["state0/" + k]) for k in ref_keys}, strict=True)
