"""oracle/ref_modules.py -- the reference's own Python modules of the hot path, as the CHECKER.  TEST INFRASTRUCTURE ONLY.

The two files that sit directly on the drop-in boundary are Python:

    depthwise_conv2d_implicit_gemm  = /root/reference/cutlass/examples/19_large_depthwise_conv2d_torch_extension/depthwise_conv2d_implicit_gemm.py
    reference_models_SLaK           = /root/reference/models/SLaK.py

/root/reference does not exist on the GPU box, and reference SOURCES are never copied into the repository.  What travels is the
same thing that travels for a compiled reference: a BUILD OUTPUT under ``oracle/_ref/`` (git-ignored, not gpurun-ignored) -- here
CPython bytecode made by ``py_compile`` from the files where they lie (``build()``, called by ``__graft_entry__.build()`` whenever
/root/reference is present).  ``load()`` returns the module from the source when the checkout is there and from the bytecode
otherwise, so ``-m gpu`` tests run the UNMODIFIED reference modules on top of ``libslak_hip.so`` on the GPU box.

What the modules need and this image lacks is supplied at the import boundary only, exactly as tests/golden/make_golden.py does
for the fixtures: ``timm.models.layers.{trunc_normal_, DropPath}`` and ``timm.models.registry.register_model``.

Round 5: the reference's step loop itself -- ``engine.py`` (``train_one_epoch``) and the ``utils.py`` it imports --
are loaded the same way (``load_engine``).  Their import boundary: ``torch._six.inf`` (removed from torch 2.x; == ``math.inf``),
``tensorboardX.SummaryWriter`` (never instantiated: log_writer=None), ``timm.data.Mixup`` / ``timm.utils.{accuracy, ModelEma,
get_state_dict}`` (type annotations and the evaluate() helper; timm1/utils/metrics.py:25-32 restated for ``accuracy``).
"""
import importlib.machinery
import importlib.util
import os
import py_compile
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "_ref")
SOURCES = {
    "depthwise_conv2d_implicit_gemm": os.path.join(REF, "cutlass", "examples", "19_large_depthwise_conv2d_torch_extension",
                                                   "depthwise_conv2d_implicit_gemm.py"),
    "reference_models_SLaK": os.path.join(REF, "models", "SLaK.py"),
    "reference_engine": os.path.join(REF, "engine.py"),          # train_one_epoch (engine.py:17-140): the step loop north_star keeps identical
    "reference_utils": os.path.join(REF, "utils.py"),            # MetricLogger / cosine_scheduler / NativeScaler, which engine.py and main.py use
}


def pyc_path(name):
    return os.path.join(OUT, name + ".pyc")


def build():
    """Compile the reference modules to bytecode under oracle/_ref/ (no-op without /root/reference)."""
    made = []
    if not os.path.isdir(REF):
        return made
    os.makedirs(OUT, exist_ok=True)
    for name, src in SOURCES.items():
        dst = pyc_path(name)
        if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            py_compile.compile(src, cfile=dst, doraise=True)
        made.append(dst)
    return made


def available(name):
    return os.path.exists(SOURCES[name]) or os.path.exists(pyc_path(name))


FORCE_BYTECODE = False        # tests set this to exercise, in the build container, the path the GPU box takes


def _exec(name, module_name):
    if os.path.exists(SOURCES[name]) and not FORCE_BYTECODE:
        spec = importlib.util.spec_from_file_location(module_name, SOURCES[name])
    else:
        loader = importlib.machinery.SourcelessFileLoader(module_name, pyc_path(name))
        spec = importlib.util.spec_from_loader(module_name, loader, origin=pyc_path(name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_dwconv_module(ext_dir):
    """The reference's depthwise_conv2d_implicit_gemm.py, unmodified, importing ``_depthwise_conv2d_implicit_gemm_C`` (:8) from
    ``ext_dir`` -- i.e. slak_amd/lib, where slak_amd.build.build_pybind() leaves the module compiled on libslak_hip.so."""
    if ext_dir not in sys.path:
        sys.path.insert(0, ext_dir)
    return _exec("depthwise_conv2d_implicit_gemm", "reference_depthwise_conv2d_implicit_gemm")


def _timm_shim():
    import torch.nn as nn
    timm = types.ModuleType("timm"); tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers"); tr = types.ModuleType("timm.models.registry")
    tl.trunc_normal_ = lambda t, mean=0., std=1., a=-2., b=2.: nn.init.trunc_normal_(t, mean, std, a, b)

    class DropPath(nn.Module):                       # timm1/layers/drop.py:137-150 (stochastic depth per sample)
        def __init__(self, drop_prob=0., scale_by_keep=True):
            super().__init__()
            self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            if keep > 0.0 and self.scale_by_keep:
                mask.div_(keep)
            return x * mask
    tl.DropPath = DropPath
    tr.register_model = lambda f: f
    return {"timm": timm, "timm.models": tm, "timm.models.layers": tl, "timm.models.registry": tr}


def load_slak_model(dwconv_module):
    """The reference's models/SLaK.py, unmodified, with ``depthwise_conv2d_implicit_gemm`` (models/SLaK.py:17) resolved to
    ``dwconv_module`` -- either the reference's own op module (load_dwconv_module) or slak_amd.depthwise_conv2d_implicit_gemm."""
    shim = _timm_shim()
    shim["depthwise_conv2d_implicit_gemm"] = dwconv_module
    saved = {k: sys.modules.get(k) for k in shim}
    sys.modules.update(shim)
    try:
        return _exec("reference_models_SLaK", "reference_models_SLaK")
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def _engine_shim():
    """What engine.py:12-15 and utils.py:15-23 import and this image lacks (names only; see the module docstring)."""
    import math
    six = types.ModuleType("torch._six"); six.inf = math.inf
    tbx = types.ModuleType("tensorboardX")

    class SummaryWriter:                             # utils.TensorboardLogger only; log_writer is None in every test
        def __init__(self, *a, **k):
            raise RuntimeError("tensorboardX is not in this image")
    tbx.SummaryWriter = SummaryWriter
    timm = types.ModuleType("timm"); td = types.ModuleType("timm.data"); tu = types.ModuleType("timm.utils")

    class Mixup:                                     # annotation only (engine.py:20); mixup_fn is None in every test
        pass

    class ModelEma:                                  # annotation only (engine.py:20): main.py:17 takes the class from model_sema.py
        pass

    def accuracy(output, target, topk=(1,)):         # timm1/utils/metrics.py:25-32
        maxk = min(max(topk), output.size()[1])
        batch_size = target.size(0)
        _, pred = output.topk(maxk, 1, True, True)
        pred = pred.t()
        correct = pred.eq(target.reshape(1, -1).expand_as(pred))
        return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / batch_size for k in topk]

    def get_state_dict(model, unwrap_fn=None):       # timm1/utils/model.py: the EMA's module state dict (utils.save_model only)
        return (unwrap_fn(model) if unwrap_fn else model).state_dict()
    td.Mixup = Mixup
    tu.accuracy, tu.ModelEma, tu.get_state_dict = accuracy, ModelEma, get_state_dict
    return {"torch._six": six, "tensorboardX": tbx, "timm": timm, "timm.data": td, "timm.utils": tu}


def load_engine():
    """(engine, utils): the reference's engine.py and utils.py, unmodified.  ``import utils`` (engine.py:15) resolves to the reference's
    utils.py loaded here; nothing stays in sys.modules afterwards."""
    shim = _engine_shim()
    saved = {k: sys.modules.get(k) for k in list(shim) + ["utils"]}
    sys.modules.update(shim)
    try:
        utils = _exec("reference_utils", "reference_utils")
        sys.modules["utils"] = utils
        engine = _exec("reference_engine", "reference_engine")
        return engine, utils
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
