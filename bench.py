#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: images/sec of a SLaK-T 51x51, 224 px, bf16 TRAIN STEP
(forward + loss + backward + AdamW step) with every depthwise conv running the hand-written HIP kernels,
plus the HBM roofline of the dominant hot-path kernel and the reference CPU nn.Conv2d path timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N --steps K --warmup W

Workload: N=1 -> BASELINE.json configs[1] (bs 128, sparsity off).  N>1 -> configs[2] (bs 128 per GPU, DDP over
RCCL, SyncBN, Masking sparsity 0.4 with prune-and-grow every 2000 steps), weak scaling.  Synthetic data
(seeded randn images, randint targets), random-init weights.  One JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def _use_tuned_gemms():
    """The pointwise (1x1) convs are plain library GEMMs (hipBLASLt through torch).  PyTorch's TunableOp picks the fastest
    hipBLASLt solution per GEMM shape; slak_amd/tuning/tunableop_gfx950.csv holds that choice for the 29 shapes of this workload,
    recorded once on an MI355X (`PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 python bench.py`, ~30 s).  It is only
    READ here (tuning off: shapes that are not in the file, or a file whose validator lines do not match this torch / ROCm /
    GPU, fall back to the default heuristic).  TunableOp reads <name><device ordinal>.csv, so every rank gets its own copy.
    SLAK_TUNED_GEMMS=0, or any PYTORCH_TUNABLEOP_* variable set by the caller, leaves everything alone."""
    src = os.path.join(ROOT, "slak_amd", "tuning", "tunableop_gfx950.csv")
    if os.environ.get("SLAK_TUNED_GEMMS", "1") == "0" or not os.path.exists(src):
        return False
    if any(k.startswith("PYTORCH_TUNABLEOP_") for k in os.environ):
        return True
    import shutil
    import tempfile
    d = tempfile.mkdtemp(prefix="slak_tunableop_")
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    shutil.copy(src, os.path.join(d, "tunableop_results%d.csv" % dev))
    os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="0", PYTORCH_TUNABLEOP_RECORD_UNTUNED="0",
                      PYTORCH_TUNABLEOP_FILENAME=os.path.join(d, "tunableop_results.csv"))
    return True


TUNED_GEMMS = _use_tuned_gemms()        # before torch is imported: TunableOp reads its environment once

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.nn.functional as F

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
STAGES_T = [(96, 56, 51, 3), (192, 28, 49, 3), (384, 14, 47, 9), (768, 7, 13, 3)]    # C, H=W, K, blocks  


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=128, help="per-GPU batch (BASELINE cfg 2/3: 128)")
    ap.add_argument("--sparsity", type=float, default=None, help="default: 0 at N=1 (cfg 2), 0.4 at N>1 (cfg 3)")
    ap.add_argument("--update-frequency", type=int, default=2000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--kernel-reps", type=int, default=30)
    ap.add_argument("--fp32-dwconv", action="store_true", help="reference dtype flow: dw convs see fp32 even under autocast")
    ap.add_argument("--cudnn-benchmark", type=int, default=1, help="torch.backends.cudnn.benchmark (main.py:235 sets it): MIOpen find mode for the stem/downsample convolutions")
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW(fused=True) (+ a separate mask-apply launch) instead of slak_amd's one-launch MaskedAdamW")
    ap.add_argument("--model-ema", action="store_true", help="also keep the reference's sparsity-aware EMA (--model_ema true recipes): one HIP launch per step")
    ap.add_argument("--no-fused-tri", action="store_true", help="run the three branch convolutions as three autograd nodes (one launch each)")
    ap.add_argument("--no-fused-bn", action="store_true", help="run the three branch BatchNorms + adds as the reference's PyTorch modules")
    ap.add_argument("--no-fused-tail", action="store_true", help="run the block tail (permute/LayerNorm/gamma/residual) as the reference's PyTorch ops")
    return ap.parse_args()


def event_time_ms(fn, reps, stream_device):
    """Average duration of fn() over `reps` launches, HIP events on the stream the kernels are launched on
    (torch's current stream == the stream slak_amd.ops passes to the C ABI)."""
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(stream_device)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def hot_path_kernels(device, batch, reps, dtype):
    """Every distinct (stage, kernel, pass) of the dw-conv hot path at the bench shapes, timed alone: the C-ABI entry points
    are called directly on preallocated buffers (the tensor-level wrappers of slak_amd.ops add ~10 us of host work per call,
    more than the smallest kernels take), HIP events on the launch stream around `reps` back-to-back launches."""
    from slak_amd import _lib, ops
    L = _lib.lib()
    st = torch.cuda.current_stream(device).cuda_stream
    out = []
    b = 2 if dtype != torch.float32 else 4
    for si, (C, HW, K, blocks) in enumerate(STAGES_T):
        x = torch.randn(batch, C, HW, HW, device=device).to(dtype)
        dy = torch.randn_like(x)
        y = torch.empty_like(x)
        dt = ops._DT[x.dtype]
        for kname, (kh, kw) in (("Kx5", (K, 5)), ("5xK", (5, K)), ("5x5", (5, 5))):
            w = torch.randn(C, 1, kh, kw, device=device) * 0.02
            dw = torch.empty_like(w)
            dims = (batch, C, HW, HW, kh, kw)
            nb = max(int(L.slak_dwconv2d_workspace_bytes(op, *dims, dt)) for op in (_lib.OP_FWD, _lib.OP_BWD_DATA, _lib.OP_BWD_FILTER))
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=device)
            a_f = (x.data_ptr(), dt, w.data_ptr(), _lib.SLAK_F32, y.data_ptr(), dt) + dims + (ws.data_ptr(), ws.numel(), st)
            a_d = (dy.data_ptr(), dt, w.data_ptr(), _lib.SLAK_F32, y.data_ptr(), dt) + dims + (ws.data_ptr(), ws.numel(), st)
            a_w = (dy.data_ptr(), dt, x.data_ptr(), dt, dw.data_ptr()) + dims + (ws.data_ptr(), ws.numel(), st)
            S = x.numel()
            for pname, fn, extra in (("fwd", lambda: _lib.check(L.slak_dwconv2d_forward(*a_f)), C * kh * kw * 4),
                                     ("bwd_data", lambda: _lib.check(L.slak_dwconv2d_backward_data(*a_d)), C * kh * kw * 4),
                                     ("bwd_filter", lambda: _lib.check(L.slak_dwconv2d_backward_filter(*a_w)), C * kh * kw * 4)):
                ms = event_time_ms(fn, reps, device)
                alg_bytes = 2 * S * b + extra                       # 2*S*b (+ C*kh*kw*4)
                out.append(dict(stage=si + 1, kernel="%dx%d" % (kh, kw), branch=kname, op=pname, ms=ms, calls_per_step=blocks,
                                alg_bytes=alg_bytes, gbs=alg_bytes / ms / 1e6, gflop_nominal=2.0 * S * kh * kw / 1e9))
            del ws, dw
        del x, dy, y
    return out


def measured_traffic(k):
    """HBM bytes per launch of kernel `k` from the committed PMC passes (profiles/pmc_traffic.json, produced by
    tools/pmc_traffic.py from two rocprofv3 --pmc runs: FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as read);
    None if that kernel/shape was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tab = json.load(f)
        e = tab.get("s%d_%s_%s" % (k["stage"], k["kernel"], k["op"]))
        return None if e is None else e["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        return None


def cpu_baseline(threads, batch=4, seconds_per_shape=1.5):
    """Reference CPU path (north star: 'the reference CPU nn.Conv2d path timed on the node's host cores'):
    torch CPU F.conv2d fp32 fwd + bwd of every distinct dw conv of SLaK-T (BASELINE cfg 1 is the stage-1 block of
    this list), weighted by how often each occurs per image -> images/s of the dw-conv hot path alone.
    Bounded sample: `batch` images per call, each of the 12 shapes repeated for ~seconds_per_shape (>= 3 calls),
    median taken -> ~20 s of CPU work in total."""
    torch.set_num_threads(threads)
    total = 0.0
    detail = {}
    ncalls = 0
    t_start = time.perf_counter()
    for (C, HW, K, blocks) in STAGES_T:
        x = torch.randn(batch, C, HW, HW, requires_grad=True)
        for kh, kw in ((K, 5), (5, K), (5, 5)):
            w = (torch.randn(C, 1, kh, kw) * 0.02).requires_grad_(True)
            def it():
                y = F.conv2d(x, w, None, 1, (kh // 2, kw // 2), 1, C)
                y.backward(torch.ones_like(y))
            it()                                                  # warm-up
            ts = []
            t_shape = time.perf_counter()
            while len(ts) < 3 or (time.perf_counter() - t_shape < seconds_per_shape and len(ts) < 200):
                t0 = time.perf_counter(); it(); ts.append(time.perf_counter() - t0)
            ncalls += len(ts)
            t = sorted(ts)[len(ts) // 2]
            detail["s%d_%dx%d" % (HW, kh, kw)] = t
            total += t * blocks
    return dict(value=batch / total, unit="images/s (dw-conv hot path only: all 54 convs fwd+bwd)", cores=threads, kind="reference",
                sample="torch %s CPU F.conv2d fp32 fwd+bwd, batch %d, every distinct SLaK-T dw-conv shape (cfg-1 stage-1 block: "
                       "%.1f ms per image), median of %d calls in total, weighted by blocks/stage; wall %.1f s" % (
                           torch.__version__, batch, 1e3 / batch * sum(v for k, v in detail.items() if k.startswith("s56")), ncalls,
                           time.perf_counter() - t_start))


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)        # "nccl" is RCCL on ROCm
    n_gpus = world if distributed else 1
    sparsity = a.sparsity if a.sparsity is not None else (0.4 if n_gpus > 1 else 0.0)

    torch.backends.cudnn.benchmark = bool(a.cudnn_benchmark)       # main.py:235
    import slak_amd.slak_model as M
    from slak_amd.sparse_core import CosineDecay, Masking
    M.Block.fused_tail = not a.no_fused_tail and not a.fp32_dwconv    # HIP glue kernels around the pointwise GEMMs
    M.ReparamLargeKernelConv.fused_bn = not a.no_fused_bn and not a.fp32_dwconv   # branch BatchNorms + adds as one HIP op
    M.LayerNorm.fused_cf = not a.no_fused_tail                     # channels_first LayerNorm of stem/downsample as one HIP kernel
    M.ReparamLargeKernelConv.fused_tri = M.ReparamLargeKernelConv.fused_bn and not a.no_fused_tri   # three branch convs as one autograd node
    from slak_amd import block_ops
    block_ops.cache_lowp_weights = True                            # bf16 weight copies refreshed by one multi-tensor launch per step
    M.use_sync_bn = True                                          # reference default (models/SLaK.py:19); falls back to BN math at world 1
    torch.manual_seed(0 + rank)                                   # main.py:232  seed = args.seed + rank
    model = M.SLaK_tiny(kernel_size=[51, 49, 47, 13, 5], Decom=True, bn=True, drop_path_rate=0.1,
                        lowp_dwconv=not a.fp32_dwconv).to(device)
    if distributed:
        model = nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=False)   # main.py:374-376
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        (no_decay if (p.dim() == 1 or n.endswith(".bias")) else decay).append(p)                                     # optim_factory.py no-decay rule
    groups = [dict(params=decay, weight_decay=0.05), dict(params=no_decay, weight_decay=0.0)]
    if a.torch_adamw:
        opt = torch.optim.AdamW(groups, lr=4e-3, fused=True)
    else:
        from slak_amd.optim_factory import MaskedAdamW               # AdamW + w *= mask + bf16 weight copies: one HIP launch
        opt = MaskedAdamW(groups, lr=4e-3)
    criterion = nn.CrossEntropyLoss(label_smoothing=0.1)
    mask = None
    if sparsity > 0:
        margs = types.SimpleNamespace(device=str(device), fix=False, update_frequency=a.update_frequency, only_L=False,
                                      sparse_init="uniform", sparsity=sparsity, distributed=distributed)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            mask = Masking(opt, None, CosineDecay(0.3, 300 * 1251), prune_rate=0.3, prune_mode="magnitude",
                           growth_mode="gradient", redistribution_mode="none", args=margs)
            mask.add_module(model)

    model_ema = None
    if a.model_ema:                                               # main.py:339-346 (--model_ema true), updated after every step: engine.py:87-88
        from slak_amd.model_sema import ModelEma
        model_ema = ModelEma(model.module if distributed else model, decay=0.9999, device='', resume='')   # built from the unwrapped model (main.py:341 precedes the DDP wrap)

    g = torch.Generator(device=device).manual_seed(1234 + rank)
    samples = torch.randn(a.batch, 3, 224, 224, device=device, generator=g)
    targets = torch.randint(0, 1000, (a.batch,), device=device, generator=g)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = criterion(model(samples), targets)
        loss.backward()
        if mask is not None:
            mask.step()                                           # engine.py:82-83
        else:
            opt.step()
        opt.zero_grad(set_to_none=True)
        if model_ema is not None:
            model_ema.update(model, mask)
        return loss

    model.train()
    for _ in range(a.warmup):
        loss = step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss.item())
    ms_per_step = 1e3 * elapsed / a.steps
    value = n_gpus * a.batch * a.steps / elapsed

    out = {
        "metric": "images/sec SLaK-T 51x51 224px bf16 train step", "value": value, "unit": "images/s",
        "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: SLaK-T 51x51 full model, bs=128, 224x224, bf16, sparsity off" if sparsity == 0 else
                                "BASELINE configs[2]: SLaK-T 51x51, bs=128/GPU, 224x224, bf16, DDP over RCCL, Masking sparsity %.2f, prune-and-grow every %d steps" % (sparsity, a.update_frequency)),
                   "global_batch": n_gpus * a.batch, "per_gpu_batch": a.batch, "parallelism": "dp%d" % n_gpus,
                   "dwconv_dtype": "fp32" if a.fp32_dwconv else "bf16 in/out, fp32 accumulate", "optimizer": "torch AdamW(fused)" if a.torch_adamw else "slak_amd MaskedAdamW (update + mask + bf16 copies, one launch)",
                   "model_ema": bool(a.model_ema),
                   "sync_bn": True, "block_tail": "hip (ln_nchw_to_nhwc + scale_residual)" if M.Block.fused_tail else "pytorch ops",
                   "branch_bn": "hip (bn3: one stats pass + one apply pass)" if M.ReparamLargeKernelConv.fused_bn else "pytorch (Sync)BatchNorm x3 + adds",
                   "pointwise_gemm": "hipBLASLt via torch" + (", TunableOp solutions from slak_amd/tuning/tunableop_gfx950.csv" if TUNED_GEMMS else ""),
                   "cudnn_benchmark": bool(a.cudnn_benchmark),
                   "final_loss": final_loss},
    }

    if rank == 0 and not a.no_roofline:
        ks = hot_path_kernels(device, a.batch, a.kernel_reps, torch.float32 if a.fp32_dwconv else torch.bfloat16)
        for k in ks:
            k["step_ms"] = k["ms"] * k["calls_per_step"]
        dom = max(ks, key=lambda k: k["step_ms"])
        hot_ms = sum(k["step_ms"] for k in ks)
        hot_bytes = sum(k["alg_bytes"] * k["calls_per_step"] for k in ks)
        out["roofline"] = {"bound": "hbm", "achieved": dom["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["gbs"] / HBM_PEAK_GBS,
                           "traffic": measured_traffic(dom), "kernel": "dwconv %s %s stage %d (N=%d)" % (dom["kernel"], dom["op"], dom["stage"], a.batch),
                           "avg_launch_ms": dom["ms"], "alg_bytes_per_launch": dom["alg_bytes"],
                           "valu_tflops_nominal": dom["gflop_nominal"] / dom["ms"]}
        out["hot_path"] = {"dwconv_ms_per_step": hot_ms, "dwconv_alg_gb_per_step": hot_bytes / 1e9,
                           "dwconv_gbs": hot_bytes / hot_ms / 1e6, "dwconv_frac_of_hbm_peak": hot_bytes / hot_ms / 1e6 / HBM_PEAK_GBS,
                           "share_of_step": hot_ms / ms_per_step,
                           "images_per_s_dwconv_only": a.batch / (hot_ms / 1e3),
                           "kernels": [{k2: (round(v, 4) if isinstance(v, float) else v) for k2, v in k.items()} for k in ks]}
    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 32))   # >32 threads only oversubscribes a 96-channel depthwise conv
    if rank == 0:
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
