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

The other BASELINE configurations run with the same command and the same JSON line:
    python bench.py --model base                      configs[3]: SLaK-B 51x51, 64 images per GPU (512 over 8), sparsity 0.4
    python bench.py --kernel 61 --res 384             configs[4]: SLaK-T 61x61 at 384 px, 64 images per GPU (256 over 4)
`mask_step` (rank 0, outside the timed region): the Masking kernels alone on the model's mask set -- apply and prune-and-grow,
HIP-event timed, against SURVEY 8(d)'s 12 / 20 bytes per masked element -- next to a CPU port of the reference step.
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
VARIANTS = {"tiny": ((3, 3, 9, 3), (96, 192, 384, 768)), "small": ((3, 3, 27, 3), (96, 192, 384, 768)),
            "base": ((3, 3, 27, 3), (128, 256, 512, 1024)), "large": ((3, 3, 27, 3), (192, 384, 768, 1536))}   # models/SLaK.py:237-280


def kernel_sizes(K):
    """[51, 49, 47, 13, 5] for K = 51 (the README recipes); K = 61 shifts the three large ones: [61, 59, 57, 13, 5]."""
    return [K, K - 2, K - 4, 13, 5]


def stages_of(model, K, res):
    """(C, H=W, K_stage, blocks) per stage: SURVEY.md Appendix A for cfg 2-5."""
    depths, dims = VARIANTS[model]
    ks = kernel_sizes(K)
    return [(dims[i], res // (4 << i), ks[i], depths[i]) for i in range(4)]


def survey_8d_bytes(stages, batch, b=2):
    """dw-conv algorithmic bytes of one train step = sum over blocks of 3 convs x 3 passes x 2*S*b (+ the filters: read forward
    and backward-data, written by backward-filter), S = batch * C * H * W.  9.88 GB for SLaK-T at 128 images (Appendix A)."""
    total = 0
    for (C, HW, K, blocks) in stages:
        S = batch * C * HW * HW
        wb = C * (2 * K * 5 + 25) * 4
        total += blocks * (9 * 2 * S * b + 3 * wb)
    return total


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--prime", type=int, default=15, help="set-up steps BEFORE the W warm-up steps (code-object loading, allocator growth, clock ramp of a cold GPU box); not part of W or K")
    ap.add_argument("--model", choices=sorted(VARIANTS), default="tiny", help="SLaK variant (BASELINE configs[3]: base)")
    ap.add_argument("--kernel", type=int, default=51, help="largest kernel: 51 -> [51,49,47,13,5]; 61 -> [61,59,57,13,5] (BASELINE configs[4])")
    ap.add_argument("--res", type=int, default=224, help="input resolution (BASELINE configs[4]: 384)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch; default 128 (configs[1], [2]), 64 for --model base (configs[3]) and --res 384 (configs[4])")
    ap.add_argument("--sparsity", type=float, default=None, help="default: 0 at N=1 (configs[1]), 0.4 at N>1 (configs[2]) and for --model base (configs[3])")
    ap.add_argument("--only-L", action="store_true", help="mask only the large-kernel LoRA weights (--only_L of the reference recipes)")
    ap.add_argument("--no-mask-bench", action="store_true", help="skip the mask_step measurement")
    ap.add_argument("--debug-mask-sync", action="store_true", help="N>1: all-reduce a mask checksum after the timed region and fail on disagreement")
    ap.add_argument("--update-frequency", type=int, default=2000)
    ap.add_argument("--graph", action="store_true", help="N = 1: forward and backward of the model replayed as two HIP graphs (torch.cuda.make_graphed_callables); "
                                                          "the loss, the optimizer / Masking step stay eager.  Takes the ~12 ms (SLaK-T) / ~25 ms (SLaK-B) of host enqueue work per step out")
    ap.add_argument("--host-profile", action="store_true", help="cProfile of the step's enqueue path, printed to stderr after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--kernel-reps", type=int, default=30)
    ap.add_argument("--fp32-dwconv", action="store_true", help="reference dtype flow: dw convs see fp32 even under autocast")
    ap.add_argument("--fp32-matrix-cores", action="store_true", help="with --fp32-dwconv: fp32 tensors through the bf16 matrix cores (opt-in two-term split, slak_set_fp32_matrix_cores)")
    ap.add_argument("--cudnn-benchmark", type=int, default=1, help="torch.backends.cudnn.benchmark (main.py:235 sets it): MIOpen find mode for the stem/downsample convolutions")
    ap.add_argument("--torch-adamw", action="store_true", help="torch.optim.AdamW(fused=True) (+ a separate mask-apply launch) instead of slak_amd's one-launch MaskedAdamW")
    ap.add_argument("--model-ema", action="store_true", help="also keep the reference's sparsity-aware EMA (--model_ema true recipes): one HIP launch per step")
    ap.add_argument("--no-fused-tri", action="store_true", help="run the three branch convolutions as three autograd nodes (one launch each)")
    ap.add_argument("--no-fused-block", action="store_true", help="keep the block's four fused ops as four autograd nodes (default: one node per block)")
    ap.add_argument("--no-fused-bn", action="store_true", help="run the three branch BatchNorms + adds as the reference's PyTorch modules")
    ap.add_argument("--no-fused-tail", action="store_true", help="run the block tail (permute/LayerNorm/gamma/residual) as the reference's PyTorch ops")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl", help="torch.distributed backend at N>1: nccl (= RCCL over xGMI) or gloo (CPU-staged; lets the N>1 code path run with several ranks on ONE GPU)")
    ap.add_argument("--device", type=int, default=None, help="GPU index of this rank (default LOCAL_RANK); --device 0 on every rank shares one GPU (gloo)")
    ap.add_argument("--markers", action="store_true", help="bracket the K timed steps with two empty marker launches (slak_debug_marker ids 1 and 2): "
                                                            "tools/step_breakdown.py cuts a rocprofv3 kernel trace exactly there")
    ap.add_argument("--dry-nccl-env", action="store_true", help="print the environment the collective library would see (NCCL_* / RCCL_* / HSA_* / rendezvous variables, as "
                                                                 "config.comm_env records them) as one JSON line and exit: no GPU work")
    ap.add_argument("--force-dist", action="store_true", help="world size 1: still init_process_group(--backend), wrap the model in DistributedDataParallel and run the fused "
                    "SyncBatchNorm exchange (all-reduces) -- the N>1 code path executes on RCCL on ONE GPU (communicator creation, DDP reducer hooks, the asynchronous "
                    "backward all-reduce); the line is NOT a scaling point (config.forced_distributed)")
    ap.add_argument("--ddp-reference-flags", action="store_true", help="N>1: DistributedDataParallel exactly as main.py:374-376 constructs it (broadcast_buffers and "
                    "gradient_as_bucket_view at their defaults); default: broadcast_buffers=False, gradient_as_bucket_view=True, the stock all-reduce comm hook (same results, less host time, one copy and 312 tiny launches less)")
    ap.add_argument("--per-step-sync", action="store_true", help="torch.cuda.synchronize() after every step, as engine.py:90 does (default: the K steps are only bracketed)")
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


def hot_path_kernels(device, batch, reps, dtype, stages, plain_too=True):
    """Every distinct launch of the dw-conv hot path at the bench shapes AS THE MODEL RUNS IT, timed alone: the C-ABI entry points
    are called directly on preallocated buffers (the tensor-level wrappers of slak_amd.ops add ~10 us of host work per call,
    more than the smallest kernels take), HIP events on the launch stream around `reps` back-to-back launches.
    Per block and pass the model runs (slak_amd.block_ops._TriDwConv):
      forward   : one launch per branch -- or ONE launch for the three (`tri`, where slak_dwconv2d_tri_supported_op(.., 0) == 1);
      bwd_data  : branch 1 plain, branches 2 and 3 ACCUMULATING into the same dx (`+acc`: autograd's adds folded in; they also read dx:
                  `alg_bytes_incl_acc_read` = 3*S*b, but the HEADLINE price stays SURVEY 8(d)'s 2*S*b per op) -- or ONE launch for the
                  three (slak_dwconv2d_tri_supported_op(.., 1) == 1);
      bwd_filter: one launch per branch -- or ONE launch for the three (where slak_dwconv2d_tri_filter_workspace_bytes > 0), or ONE for
                  the K x 5 and the 5 x 5 branch (`pair`, where slak_dwconv2d_pair_filter_workspace_bytes > 0) beside the 5 x K launch;
      bwd_data+filter: ONE launch for the data gradient and the three weight gradients (where slak_dwconv2d_tri_backward_supported == 1).
    Algorithmic bytes: SURVEY.md 8(d) per op -- 2*S*b (+ C*kh*kw*4); a three-branch (two-branch) launch is priced at the per-op figure
    of the three (two) ops it replaces (3 x 2*S*b, 2 x 2*S*b), as 8(d) prescribes."""
    import ctypes
    from slak_amd import _lib, ops, block_ops
    L = _lib.lib()
    st = torch.cuda.current_stream(device).cuda_stream
    out = []
    b = 2 if dtype != torch.float32 else 4
    stats_in_conv = bool(block_ops.bn_stats_in_conv)
    # the GPU has idled through the CPU baselines by now: ~0.3 s of work first, as the --prime steps do for the training loop (clocks ramp up
    # slowly: the first launches after an idle period measured 10-15 % slower than the same launches in tools/time_all.py)
    wa = torch.randn(4096, 4096, device=device, dtype=torch.bfloat16)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.3:
        for _ in range(20):
            wa @ wa
        torch.cuda.synchronize(device)
    del wa
    for si, (C, HW, K, blocks) in enumerate(stages):
        x = torch.randn(batch, C, HW, HW, device=device).to(dtype)
        dys = [torch.randn_like(x) for _ in range(3)]
        ys = [torch.empty_like(x) for _ in range(3)]
        dt = ops._DT[x.dtype]
        S = x.numel()
        shapes = (("Kx5", (K, 5)), ("5xK", (5, K)), ("5x5", (5, 5)))
        wts = [torch.randn(C, 1, kh, kw, device=device) * 0.02 for _, (kh, kw) in shapes]
        tri_f = dtype != torch.float32 and L.slak_dwconv2d_tri_supported_op(dt, batch, C, HW, HW, K, 0) == 1
        tri_d = dtype != torch.float32 and L.slak_dwconv2d_tri_supported_op(dt, batch, C, HW, HW, K, 1) == 1

        def add(kernel, branch, op, fn, alg_bytes, flop, extra_read=0):
            ms = event_time_ms(fn, reps, device)
            name = L.slak_debug_last_kernel()
            out.append(dict(stage=si + 1, kernel=kernel, branch=branch, op=op, ms=ms, calls_per_step=blocks, alg_bytes=alg_bytes,
                            alg_bytes_incl_acc_read=alg_bytes + extra_read, gbs=alg_bytes / ms / 1e6, gflop_nominal=flop / 1e9,
                            hip_kernel=name.decode() if name else ""))
        wbytes = sum(C * kh * kw * 4 for _, (kh, kw) in shapes)
        flops3 = sum(2.0 * S * kh * kw for _, (kh, kw) in shapes)
        tri_w_nb = int(L.slak_dwconv2d_tri_filter_workspace_bytes(dt, batch, C, HW, HW, K)) if dtype != torch.float32 else 0
        # the data gradient and the three weight gradients in ONE launch (the 14 x 14 class; block_ops._tri_backward_impl takes it where it exists):
        # priced at the per-op figures of the six ops it replaces
        tri_b = bool(tri_w_nb) and block_ops.fused_tri_backward and block_ops.fused_tri_wgrad and L.slak_dwconv2d_tri_backward_supported(dt, batch, C, HW, HW, K) == 1
        if tri_b:
            dws3 = [torch.empty_like(w) for w in wts]
            ws3 = torch.empty(tri_w_nb, dtype=torch.uint8, device=device)
            a_tb = (dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), wts[0].data_ptr(), wts[1].data_ptr(), wts[2].data_ptr(), ys[0].data_ptr(),
                    dws3[0].data_ptr(), dws3[1].data_ptr(), dws3[2].data_ptr(), dt, batch, C, HW, HW, K, ws3.data_ptr(), tri_w_nb, st)
            add("%dx5+5x%d+5x5" % (K, K), "tri", "bwd_data+filter", lambda: _lib.check(L.slak_dwconv2d_tri_backward(*a_tb)), 2 * (3 * 2 * S * b + wbytes), 2 * flops3)
            out[-1]["variant"] = "data gradient + three weight gradients in one launch (slak_dwconv2d_tri_backward: what the training step launches)"
        elif tri_w_nb:
            dws3 = [torch.empty_like(w) for w in wts]
            ws3 = torch.empty(tri_w_nb, dtype=torch.uint8, device=device)
            a_tw = (dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dws3[0].data_ptr(), dws3[1].data_ptr(), dws3[2].data_ptr(),
                    dt, batch, C, HW, HW, K, ws3.data_ptr(), tri_w_nb, st)
            add("%dx5+5x%d+5x5" % (K, K), "tri", "bwd_filter", lambda: _lib.check(L.slak_dwconv2d_tri_backward_filter(*a_tw)), 3 * 2 * S * b + wbytes, flops3)
        if tri_f or tri_d:
            a_tf = (x.data_ptr(), wts[0].data_ptr(), wts[1].data_ptr(), wts[2].data_ptr(), ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(), dt, batch, C, HW, HW, K, st)
            a_td = (dys[0].data_ptr(), dys[1].data_ptr(), dys[2].data_ptr(), wts[0].data_ptr(), wts[1].data_ptr(), wts[2].data_ptr(), ys[0].data_ptr(), dt, batch, C, HW, HW, K, st)
            if tri_f:
                # the model (block_ops._TriDwConv, training) launches the variant that also gathers the branch BatchNorms' batch sums
                # wherever it exists (SLAK_BN_STATS_IN_CONV, default on): THAT launch is timed; the plain one is kept beside it
                rows = int(L.slak_dwconv2d_tri_stats_rows(dt, batch, C, HW, HW, K)) if stats_in_conv else 0
                if rows > 0:
                    stt = torch.empty((rows, C, 6), dtype=torch.float32, device=device)
                    a_ts = a_tf[:7] + (stt.data_ptr(),) + a_tf[7:]
                    add("%dx5+5x%d+5x5" % (K, K), "tri", "fwd", lambda: _lib.check(L.slak_dwconv2d_tri_forward_stats(*a_ts)), 3 * 2 * S * b + wbytes, flops3)
                    out[-1]["variant"] = "forward + BatchNorm batch sums (slak_dwconv2d_tri_forward_stats: what the training step launches)"
                    if plain_too:                                  # (tools/time_all.py under a profiler: one dispatch group per entry)
                        out[-1]["plain_forward_ms"] = event_time_ms(lambda: _lib.check(L.slak_dwconv2d_tri_forward(*a_tf)), reps, device)
                else:
                    add("%dx5+5x%d+5x5" % (K, K), "tri", "fwd", lambda: _lib.check(L.slak_dwconv2d_tri_forward(*a_tf)), 3 * 2 * S * b + wbytes, flops3)
            if tri_d and not tri_b:
                add("%dx5+5x%d+5x5" % (K, K), "tri", "bwd_data", lambda: _lib.check(L.slak_dwconv2d_tri_backward_data(*a_td)), 3 * 2 * S * b + wbytes, flops3)
        pair_nb = int(L.slak_dwconv2d_pair_filter_workspace_bytes(dt, batch, C, HW, HW, K)) if (dtype != torch.float32 and not tri_w_nb) else 0
        if pair_nb:
            dwp = [torch.empty_like(wts[0]), torch.empty_like(wts[2])]
            wsp = torch.empty(pair_nb, dtype=torch.uint8, device=device)
            a_pw = (dys[0].data_ptr(), dys[2].data_ptr(), x.data_ptr(), dwp[0].data_ptr(), dwp[1].data_ptr(), dt, batch, C, HW, HW, K, wsp.data_ptr(), pair_nb, st)
            if L.slak_dwconv2d_pair_backward_filter(*a_pw) == _lib.OK:
                add("%dx5+5x5" % K, "pair", "bwd_filter", lambda: _lib.check(L.slak_dwconv2d_pair_backward_filter(*a_pw)),
                    2 * 2 * S * b + C * (K * 5 + 25) * 4, 2.0 * S * (K * 5 + 25))
            else:
                pair_nb = 0
        for bi, (kname, (kh, kw)) in enumerate(shapes):
            w = wts[bi]
            dw = torch.empty_like(w)
            dims = (batch, C, HW, HW, kh, kw)
            nb = max(int(L.slak_dwconv2d_workspace_bytes(op, *dims, dt)) for op in (_lib.OP_FWD, _lib.OP_BWD_DATA, _lib.OP_BWD_FILTER))
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=device)
            a_f = (x.data_ptr(), dt, w.data_ptr(), _lib.SLAK_F32, ys[0].data_ptr(), dt) + dims + (ws.data_ptr(), ws.numel(), st)
            a_d = (dys[0].data_ptr(), dt, w.data_ptr(), _lib.SLAK_F32, ys[0].data_ptr(), dt) + dims + (ws.data_ptr(), ws.numel(), st)
            a_w = (dys[0].data_ptr(), dt, x.data_ptr(), dt, dw.data_ptr()) + dims + (ws.data_ptr(), ws.numel(), st)
            kn, flop, wb = "%dx%d" % (kh, kw), 2.0 * S * kh * kw, C * kh * kw * 4
            if not tri_f:
                st_ok = False
                if stats_in_conv and dtype == torch.bfloat16:      # per-branch launches that gather their BatchNorm's sums (ops.dwconv2d_forward_stats)
                    stb = torch.empty((4 * batch, C, 2), dtype=torch.float32, device=device)
                    rws = ctypes.c_int(0)
                    a_fs = (x.data_ptr(), dt, w.data_ptr(), _lib.SLAK_F32, ys[0].data_ptr(), dt, stb.data_ptr(), 4 * batch, ctypes.byref(rws)) + dims + (st,)
                    st_ok = L.slak_dwconv2d_forward_stats(*a_fs) == _lib.OK
                if st_ok:
                    add(kn, kname, "fwd", lambda: _lib.check(L.slak_dwconv2d_forward_stats(*a_fs)), 2 * S * b + wb, flop)
                    out[-1]["variant"] = "forward + BatchNorm batch sums (slak_dwconv2d_forward_stats: what the training step launches)"
                else:
                    add(kn, kname, "fwd", lambda: _lib.check(L.slak_dwconv2d_forward(*a_f)), 2 * S * b + wb, flop)
            if not tri_d:
                acc_ok = bi > 0 and dtype != torch.float32 and L.slak_dwconv2d_backward_data_accumulate(*a_d) == _lib.OK
                if acc_ok:
                    add(kn, kname, "bwd_data+acc", lambda: _lib.check(L.slak_dwconv2d_backward_data_accumulate(*a_d)), 2 * S * b + wb, flop, extra_read=S * b)
                else:
                    add(kn, kname, "bwd_data", lambda: _lib.check(L.slak_dwconv2d_backward_data(*a_d)), 2 * S * b + wb, flop)
            if not tri_w_nb and not (pair_nb and bi != 1):
                add(kn, kname, "bwd_filter", lambda: _lib.check(L.slak_dwconv2d_backward_filter(*a_w)), 2 * S * b + wb, flop)
            del ws, dw
        del x, dys, ys
    return out


def measured_traffic(k):
    """HBM bytes per launch of kernel `k` from the committed PMC passes (profiles/pmc_traffic.json, produced by
    tools/pmc_traffic.py from two rocprofv3 --pmc runs: FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as read);
    None if that kernel/shape was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            tab = json.load(f)
        e = tab.get("s%d_%s_%s" % (k["stage"], k["kernel"], k["op"]))
        return None if (e is None or e.get("alg_bytes") != k["alg_bytes"]) else e["hbm_bytes_per_launch"]   # same shape only
    except (OSError, ValueError):
        return None


def cpu_baseline(threads, stages, label, batch=4, seconds_per_shape=1.5):
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
    if stages[0][1] > 64:
        batch = 2                                                 # 96x96 planes: keep the sample inside ~20-30 s
    for (C, HW, K, blocks) in stages:
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
    s1 = "s%d_" % stages[0][1]
    return dict(value=batch / total, unit="images/s (dw-conv hot path only: all %d convs fwd+bwd)" % (3 * sum(st[3] for st in stages)),
                cores=threads, kind="reference",
                sample="torch %s CPU F.conv2d fp32 fwd+bwd, batch %d, every distinct %s dw-conv shape (stage-1 block: "
                       "%.1f ms per image), median of %d calls in total, weighted by blocks/stage; wall %.1f s" % (
                           torch.__version__, batch, label, 1e3 / batch * sum(v for k, v in detail.items() if k.startswith(s1)), ncalls,
                           time.perf_counter() - t_start))


def mask_reference_cpu(ws, ms, gs, rate, elems, gpu_update_ms, budget_s=15.0):
    """The reference's prune-and-grow step with the reference's own algorithm -- full torch.sort per tensor, twice -- on the host cores
    (the thread count the dw-conv CPU baseline uses -- min(32, cores): 256 threads on a torch.sort of a 120 K-element tensor only
    oversubscribe, measured 2x slower), on a bounded sample of the mask set of the bench model:
      magnitude_prune (funcs.py:107-114): num_remove = ceil(rate * nnz), k = ceil(zeros + num_remove), idx = sort(|w|) -> mask[idx[:k]] = 0
      gradient_growth (funcs.py:196-205): g = grad * (mask == 0), idx = sort(|g|, descending) -> mask[idx[:removed]] = 1
      apply_mask      (sparse_core.py:316-333): w = w * mask
    Per tensor: prune, grow, apply (truncate_weights, sparse_core.py:335-357, runs the three as separate loops over the tensors: the
    same work per tensor)."""
    import math
    threads = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(threads)
    cw = [w.detach().cpu().clone() for w in ws]; cm = [m.detach().cpu().clone() for m in ms]; cg = [g.detach().cpu() for g in gs]
    def update_one(w, m, g):
        nnz = int(m.sum().item()); zeros = m.numel() - nnz
        num_remove = math.ceil(rate * nnz)
        k = math.ceil(zeros + num_remove)
        _, idx = torch.sort(torch.abs(w.flatten()))
        m.view(-1)[idx[:k]] = 0.0
        gg = g * (m == 0).float()
        _, idx = torch.sort(torch.abs(gg).flatten(), descending=True)
        m.view(-1)[idx[:num_remove]] = 1.0
        w.mul_(m)
    # a BOUNDED sample: tensors in model order until the budget is spent (the whole SLaK-T set takes this algorithm > 2 minutes on
    # 256 threads); the per-update figure is the sample's rate applied to the full set
    t0 = time.perf_counter(); done_t = done_e = 0
    for w, m, g in zip(cw, cm, cg):
        update_one(w, m, g)
        done_t += 1; done_e += w.numel()
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    rate_meps = done_e / dt / 1e6
    full_ms = elems / rate_meps / 1e3
    return {"kind": "reference", "cores": threads, "ms_per_update": full_ms, "elements": elems, "tensors": len(ws),
            "sample_tensors": done_t, "sample_elements": done_e, "sample_seconds": round(dt, 2),
            "melem_per_s": rate_meps, "gpu_over_cpu": full_ms / gpu_update_ms,
            "sample": "the reference algorithm (funcs.py:107-114 magnitude_prune, :196-205 gradient_growth: a full torch.sort per tensor and "
                      "direction, then sparse_core.py:316-333 w *= mask) in torch %s on the host, torch.set_num_threads(%d), the first %d of %d "
                      "tensors in model order (%d of %d elements, %.1f s); ms_per_update = that rate applied to the full set"
                      % (torch.__version__, threads, done_t, len(ws), done_e, elems, dt)}


def mask_step_bench(device, model_name, ks, only_L, reps=20, cpu_seconds=6.0):
    """The Masking kernels alone: `slak_mask_apply` (ordinary steps when the optimizer does not fold
    the mask in; 12 B per masked element: w r/w + mask r) and `slak_mask_prune_and_grow` (update steps; fused ideal 20 B per element:
    w r/w 8 + grad r 4 + mask r/w 8) over the mask set of the bench model -- every 2-D / 4-D parameter (95 tensors / 30.7 M elements
    for SLaK-T; --only-L: the 36 LoRA tensors) -- through the C ABI, HIP events on the launch stream.  `passes` = how many times the
    implementation streams the tensors (mask_kernels.hip: histogram + compaction + membership per k-th-element search, the prune
    membership pass also being the regrow histogram pass and the regrow membership pass also being the apply, the pruned mask travelling
    between them as a byte per four elements: 5 passes, 41 B/elem).  CPU side: the numpy port of sparse_core.Masking.truncate_weights / funcs.magnitude_prune /
    gradient_growth (oracle/mask_oracle.py; the reference itself is not on this box) on a bounded prefix of the same tensors."""
    import ctypes
    from slak_amd import _lib
    from slak_amd.slak_model import slak_mask_set_shapes
    L = _lib.lib()
    st = torch.cuda.current_stream(device).cuda_stream
    shapes = slak_mask_set_shapes(model_name, tuple(ks), only_L=only_L)
    g = torch.Generator(device=device).manual_seed(99)
    ws = [torch.randn(s, device=device, generator=g) * 0.02 for s in shapes]
    ms = [(torch.rand(s, device=device, generator=g) < 0.6).float() for s in shapes]
    for w, m in zip(ws, ms):
        w.mul_(m)
    gs = [torch.randn(s, device=device, generator=g) for s in shapes]
    segs = (_lib.MaskSegment * len(shapes))()
    for i in range(len(shapes)):
        segs[i].weight, segs[i].mask, segs[i].grad, segs[i].momentum, segs[i].numel = ws[i].data_ptr(), ms[i].data_ptr(), gs[i].data_ptr(), None, ws[i].numel()
    plan = ctypes.c_void_p()
    _lib.check(L.slak_mask_plan_create(segs, len(shapes), ctypes.byref(plan)), "slak_mask_plan_create")
    elems = sum(w.numel() for w in ws)
    t_apply = event_time_ms(lambda: _lib.check(L.slak_mask_apply(plan, st)), reps, device)
    # every timed prune-and-grow starts from the same state (masked trained weights, density 0.6): repeating it on its own output would
    # prune weights that were regrown at exactly 0 a moment ago, which no training run does (2000 optimizer steps lie between updates)
    ws0 = [w.clone() for w in ws]; ms0 = [m.clone() for m in ms]
    times = []
    for rep in range(reps + 3):
        torch._foreach_copy_(ws, ws0); torch._foreach_copy_(ms, ms0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(L.slak_mask_prune_and_grow(plan, 0.3, st))
        e1.record(); e1.synchronize()
        if rep >= 3:
            times.append(e0.elapsed_time(e1))
    t_update = float(sorted(times)[len(times) // 2])              # median
    dens_m = [m.clone() for m in ms]                             # the masks one update leaves
    torch._foreach_copy_(ws, ws0); torch._foreach_copy_(ms, ms0)  # the CPU legs below start from the same state
    del ws0, ms0
    density = float(sum(m.sum().item() for m in dens_m)) / elems
    del dens_m
    L.slak_mask_plan_destroy(plan)
    out = {"tensors": len(shapes), "elements": elems, "only_L": bool(only_L),
           "apply_ms": t_apply, "apply_alg_bytes": 12 * elems, "apply_gbs": 12 * elems / t_apply / 1e6, "apply_frac_of_hbm_peak": 12 * elems / t_apply / 1e6 / HBM_PEAK_GBS,
           "update_ms": t_update, "update_alg_bytes": 20 * elems, "update_gbs": 20 * elems / t_update / 1e6, "update_frac_of_hbm_peak": 20 * elems / t_update / 1e6 / HBM_PEAK_GBS,
           "update_passes_over_keys": 5, "update_launches": 13, "update_bytes_moved_per_elem": 41, "density_after": density,
           "note": "apply is folded into MaskedAdamW's update on ordinary steps (0 extra bytes); prune-and-grow runs every update_frequency steps"}
    # CPU port on a bounded prefix of the same mask set
    import oracle
    t0 = time.perf_counter(); done = 0; n_t = 0
    order = sorted(range(len(shapes)), key=lambda i: ws[i].numel())          # small tensors first: several tensors inside the budget
    cw, cm, cg = {}, {}, {}
    for i in order:
        cw[str(i)] = ws[i].cpu().numpy(); cm[str(i)] = ms[i].cpu().numpy(); cg[str(i)] = gs[i].cpu().numpy()
        done += ws[i].numel(); n_t += 1
        if done >= 4_000_000:
            break
    t1 = time.perf_counter(); reps_cpu = 0
    while reps_cpu < 1 or (time.perf_counter() - t1 < cpu_seconds and reps_cpu < 20):
        oracle.truncate_weights(cw, cm, cg, 0.3); reps_cpu += 1
    dt = (time.perf_counter() - t1) / reps_cpu
    out["cpu_reference"] = mask_reference_cpu(ws, ms, gs, 0.3, elems, t_update)
    out["cpu_port"] = {"kind": "port", "cores": 1, "ms_per_update": dt * 1e3, "elements": done, "tensors": n_t,
                       "melem_per_s": done / dt / 1e6, "gpu_melem_per_s": elems / (t_update / 1e3) / 1e6,
                       "sample": "oracle/mask_oracle.py truncate_weights (numpy restatement of sparse_core.py:335-357, funcs.py:107-114,196-205) on the %d smallest tensors (%d elements), %d repetitions" % (n_t, done, reps_cpu)}
    return out


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dry_nccl_env:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        print(json.dumps({"comm_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_VISIBLE", "ROCR_VISIBLE", "CUDA_VISIBLE", "TORCH_NCCL", "MASTER_", "GLOO_"))
                                       or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OMP_NUM_THREADS")},
                          "world_size": world, "rank": rank, "backend": a.backend, "torch": torch.__version__,
                          "rccl_available": bool(dist.is_available() and dist.is_nccl_available())}))
        return
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no CPU fallback for the hot path)")
    dev_index = local_rank if a.device is None else a.device
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    distributed = world > 1 or a.force_dist
    if a.force_dist and world == 1:
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("SLAK_FORCE_BN_EXCHANGE", "1")        # block_ops._bn3_group: the SyncBatchNorm exchange at world size 1 too (0: DDP's part alone, for the cost split)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)    # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")
    n_gpus = world if distributed else 1
    ks = kernel_sizes(a.kernel)
    stages = stages_of(a.model, a.kernel, a.res)
    if a.fp32_matrix_cores:
        from slak_amd import ops as _ops
        _ops.allow_fp32_matrix_cores(True)
    batch = a.batch if a.batch is not None else (64 if (a.model != "tiny" or a.res > 224) else 128)
    a.batch = batch
    sparsity = a.sparsity if a.sparsity is not None else (0.4 if (n_gpus > 1 or a.model == "base") and a.res == 224 else 0.0)
    label = "SLaK-%s %dx%d" % (a.model[0].upper(), a.kernel, a.kernel)

    torch.backends.cudnn.benchmark = bool(a.cudnn_benchmark)       # main.py:235
    import slak_amd.slak_model as M
    from slak_amd.sparse_core import CosineDecay, Masking
    M.Block.fused_tail = not a.no_fused_tail and not a.fp32_dwconv    # HIP glue kernels around the pointwise GEMMs
    M.ReparamLargeKernelConv.fused_bn = not a.no_fused_bn and not a.fp32_dwconv   # branch BatchNorms + adds as one HIP op
    M.LayerNorm.fused_cf = not a.no_fused_tail                     # channels_first LayerNorm of stem/downsample as one HIP kernel
    M.SLaK.fused_stem = os.environ.get("SLAK_FUSED_STEM", "1") != "0"
    M.SLaK.fused_downsample = not a.no_fused_tail and os.environ.get("SLAK_FUSED_DOWNSAMPLE", "1") != "0"   # LN -> 2x2/s2 conv as LN-to-patch kernel + library GEMMs
    M.ReparamLargeKernelConv.fused_tri = M.ReparamLargeKernelConv.fused_bn and not a.no_fused_tri   # three branch convs as one autograd node
    M.Block.fused_block = (M.Block.fused_tail and M.ReparamLargeKernelConv.fused_tri and not a.no_fused_block
                           and os.environ.get("SLAK_FUSED_BLOCK", "1") != "0")          # the whole block as ONE autograd node (same launches)
    from slak_amd import block_ops
    block_ops.cache_lowp_weights = True                            # bf16 weight copies refreshed by one multi-tensor launch per step
    M.use_sync_bn = True                                          # reference default (models/SLaK.py:19); falls back to BN math at world 1
    torch.manual_seed(0 + rank)                                   # main.py:232  seed = args.seed + rank
    drop_path = {"tiny": 0.1, "small": 0.4, "base": 0.5, "large": 0.5}[a.model]      # README training recipes
    model = M.create_model("SLaK_" + a.model, kernel_size=ks, Decom=True, bn=True, drop_path_rate=drop_path,
                           lowp_dwconv=not a.fp32_dwconv).to(device)
    if distributed:
        # main.py:374-376 builds DistributedDataParallel(model, device_ids=[args.gpu], find_unused_parameters=False).  The two other constructor flags are
        # results-neutral here and set for speed (--ddp-reference-flags restores the constructor's defaults): broadcast_buffers=False -- the only buffers are
        # the (Sync)BatchNorm running statistics and counters, rank-identical by construction (the statistics are all-reduced), so the per-forward broadcast from
        # rank 0 (a collective + ~1.7 ms of host time per step) re-sends what every rank already holds; gradient_as_bucket_view=True -- gradients are views of
        # the all-reduce buckets instead of being copied into them (123 MB per step).
        from slak_amd import ddp as slak_ddp
        # reference flags: the constructor call as it is.  Default: + broadcast_buffers=False, gradient_as_bucket_view=True, one collective per bucket with the
        # division inside (ncclAvg; the stock per-bucket hook on gloo), block gradients written into the bucket views (slak_amd/ddp.py)
        model = slak_ddp.wrap(model, [dev_index], reference_flags=a.ddp_reference_flags)
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
        margs = types.SimpleNamespace(device=str(device), fix=False, update_frequency=a.update_frequency, only_L=a.only_L,
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
    samples = torch.randn(a.batch, 3, a.res, a.res, device=device, generator=g)
    targets = torch.randint(0, 1000, (a.batch,), device=device, generator=g)

    fwd = model
    if a.graph:
        if distributed:
            sys.exit("--graph is wired for N = 1 (under DDP the graphed callable has to be made before the wrap)")
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=False):
            fwd = torch.cuda.make_graphed_callables(model, (samples,), num_warmup_iters=3)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16, cache_enabled=not a.graph):
            loss = criterion(fwd(samples), targets)
        loss.backward()
        if mask is not None:
            mask.step()                                           # engine.py:82-83
        else:
            opt.step()
        opt.zero_grad(set_to_none=True)
        if model_ema is not None:
            model_ema.update(model, mask)
        if a.per_step_sync:
            torch.cuda.synchronize()                              # engine.py:90
        return loss

    model.train()
    for _ in range(a.prime):                                      # a fresh box: the first ~0.3 s of work run with cold code objects, allocator and clocks
        step()
    for _ in range(a.warmup):
        loss = step()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    if a.markers:
        from slak_amd import _lib as _mk
        _mk.check(_mk.lib().slak_debug_marker(1, torch.cuda.current_stream(device).cuda_stream), "slak_debug_marker")
    host_s = 0.0                                                  # time the host spends ENQUEUEING the steps (no sync inside)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        h0 = time.perf_counter()
        loss = step()
        host_s += time.perf_counter() - h0
    if a.markers:
        _mk.check(_mk.lib().slak_debug_marker(2, torch.cuda.current_stream(device).cuda_stream), "slak_debug_marker")
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # The K timed steps above enqueue faster than the GPU executes: after a few steps the runtime's launch queue is full and every further launch
    # WAITS for the GPU, so `host_s` tends to the GPU's step time whatever the host costs.  The host's own cost per step: three steps enqueued
    # on a drained queue (nothing to wait for), synchronised afterwards.
    host_free_s = float("inf")                                    # (three rounds, the quietest one: a drained-queue round of 3 steps is easily disturbed)
    for _ in range(3):
        h0 = time.perf_counter()
        for _ in range(3):
            step()
        host_free_s = min(host_free_s, (time.perf_counter() - h0) / 3)
        torch.cuda.synchronize()
    if a.host_profile:                                            # after the timed region: where the enqueue time goes
        import cProfile, pstats
        pr = cProfile.Profile()
        for i in range(a.steps):                                  # (three steps at a time on a drained queue: the host's own cost, see above)
            pr.enable(); step(); pr.disable()
            if i % 3 == 2:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(60)
    per_rank_ms = [1e3 * elapsed / a.steps]
    mask_sync = None
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        per_rank_ms = [1e3 * x.item() / a.steps for x in allt]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
        if mask is not None:                                      # outside the timed region: one 24-byte all-reduce
            agree = mask.ranks_agree()
            mask_sync = {"ranks_agree": agree, "resyncs": mask.rank_resyncs}
            if a.debug_mask_sync and not agree:
                sys.exit("mask checksum differs across ranks")
    final_loss = float(loss.item())
    ms_per_step = 1e3 * elapsed / a.steps
    value = n_gpus * a.batch * a.steps / elapsed

    if a.model == "tiny" and a.kernel == 51 and a.res == 224:
        workload = ("BASELINE configs[1]: SLaK-T 51x51 full model, bs=128, 224x224, bf16, sparsity off" if sparsity == 0 else
                    "BASELINE configs[2]: SLaK-T 51x51, bs=%d/GPU, 224x224, bf16, DDP over RCCL, Masking sparsity %.2f, prune-and-grow every %d steps" % (a.batch, sparsity, a.update_frequency))
    elif a.model == "base" and a.kernel == 51 and a.res == 224:
        workload = "BASELINE configs[3]: SLaK-B 51x51, bs=%d/GPU (512 over 8 GPUs), 224x224, bf16, Masking sparsity %.2f, prune-and-grow every %d steps" % (a.batch, sparsity, a.update_frequency)
    elif a.model == "tiny" and a.kernel == 61 and a.res == 384:
        workload = "BASELINE configs[4]: SLaK-T 61x61, bs=%d/GPU (256 over 4 GPUs), 384x384 input, bf16, sparsity %s" % (a.batch, "off" if sparsity == 0 else "%.2f" % sparsity)
    else:
        workload = "%s, bs=%d/GPU, %dx%d, bf16, sparsity %.2f (not a BASELINE configuration)" % (label, a.batch, a.res, a.res, sparsity)
    tun = None
    try:
        tun = {"enabled": bool(torch.cuda.tunable.is_enabled()), "solutions_in_use": len(torch.cuda.tunable.get_results())}
    except Exception:
        pass
    out = {
        "metric": "images/sec %s %dpx bf16 train step" % (label, a.res), "value": value, "unit": "images/s",
        "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": workload,
                   "global_batch": n_gpus * a.batch, "per_gpu_batch": a.batch, "parallelism": "dp%d" % n_gpus,
                   "kernel_sizes": ks, "resolution": a.res, "variant": a.model,
                   "dwconv_dtype": ("fp32" + (" on the bf16 matrix cores (two-term split, three MFMAs per product)"
                                              if (a.fp32_matrix_cores or os.environ.get("SLAK_FP32_AUTOCAST_SPLIT", "0") == "1") else " (exact VALU kernels)")) if a.fp32_dwconv else "bf16 in/out, fp32 accumulate", "optimizer": "torch AdamW(fused)" if a.torch_adamw else "slak_amd MaskedAdamW (update + mask + bf16 copies, one launch)",
                   "model_ema": bool(a.model_ema), "one_autograd_node_per_block": bool(M.Block.fused_block),
                   "block_runner": bool(M.Block.fused_block and block_ops._runner() is not None),   # the blocks' call sequences issued from C++ (round 5: under DDP / SyncBatchNorm too)
                   "forced_distributed": bool(a.force_dist and world == 1),
                   "ddp": (None if not distributed else ("main.py:374-376 defaults" if a.ddp_reference_flags else "broadcast_buffers=False, gradient_as_bucket_view=True, " + getattr(model, "_slak_comm_hook", "?")
                                                          + (", block gradients written into the bucket views (no per-parameter copy: %d destinations used)" % block_ops.grad_slot_hits if block_ops.grad_slots_enabled else ""))),
                   "sync_bn_backward_exchange": (None if not distributed else ("own stream, overlapped with the pointwise weight gradients" if block_ops._bn_bwd_async else "on the compute stream")),
                   "sync_bn": True, "block_tail": "hip (ln_nchw_to_nhwc + scale_residual)" if M.Block.fused_tail else "pytorch ops",
                   "branch_bn": "hip (bn3: one stats pass + one apply pass)" if M.ReparamLargeKernelConv.fused_bn else "pytorch (Sync)BatchNorm x3 + adds",
                   "pointwise_gemm": "slak_linear_gemm / slak_linear_nt / slak_linear_wgrad (own kernels: pwconv1 + GELU, dz W2 + GELU', the stage-1 products, every weight gradient); hipBLASLt via torch for pwconv2 / dy1 W1 of stages 2-4 and the downsample products" + (", TunableOp solutions from slak_amd/tuning/tunableop_gfx950.csv" if TUNED_GEMMS else ""),
                   "tunableop": tun,
                   "cudnn_benchmark": bool(a.cudnn_benchmark),
                   "prime_steps": a.prime,
                   "world_size": (dist.get_world_size() if distributed else 1),
                   "backend": (a.backend if distributed else None),
                   "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if (distributed and a.backend == "nccl") else None),
                   # what the collective library will see (the first real 8-GPU run must be diagnosable from its JSON line alone)
                   "comm_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_VISIBLE", "ROCR_VISIBLE", "CUDA_VISIBLE", "TORCH_NCCL", "MASTER_", "GLOO_"))
                                or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "OMP_NUM_THREADS")},
                   "per_step_sync": bool(a.per_step_sync),
                   "hip_graph": ("forward and backward of the model replayed as two captured HIP graphs (torch.cuda.make_graphed_callables)" if a.graph else None),
                   "timing": "K steps between barrier + torch.cuda.synchronize() on both sides; " + ("a synchronize after every step as engine.py:90" if a.per_step_sync else "no synchronize inside (GPU-busy time == step time: the host runs ahead)"),
                   "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms],
                   "host_enqueue_ms_per_step": round(1e3 * host_free_s, 3),
                   "host_enqueue_ms_per_step_in_timed_region": round(1e3 * host_s / a.steps, 3),
                   "mask_sync": mask_sync,
                   "final_loss": final_loss},
    }

    if rank == 0 and not a.no_roofline:
        del samples
        torch.cuda.empty_cache()
        kl = hot_path_kernels(device, a.batch, a.kernel_reps, torch.float32 if a.fp32_dwconv else torch.bfloat16, stages)
        for k in kl:
            k["step_ms"] = k["ms"] * k["calls_per_step"]
        # the dominant launch: the stage-1 launches and the fused stage-3 backward cost within a few per cent of each other per step, so the
        # plain maximum flips from run to run -- among those within 10 % of the largest (round 6: 5 % let the stage-1 launches drop out of the
        # set by a hair on some boxes and the line then quoted the stage-3 launch's 0.81), the one FURTHEST from the roofline is reported
        top = max(k["step_ms"] for k in kl)
        near = [k for k in kl if k["step_ms"] >= 0.90 * top]
        dom = min(near, key=lambda k: k["gbs"])
        hot_ms = sum(k["step_ms"] for k in kl)
        hot_bytes = sum(k["alg_bytes"] * k["calls_per_step"] for k in kl)               # 2*S*b per op and pass (9.88 GB for SLaK-T)
        hot_bytes_acc = sum(k["alg_bytes_incl_acc_read"] * k["calls_per_step"] for k in kl)
        pmc = [(measured_traffic(k), k["calls_per_step"]) for k in kl]
        hot_bytes_pmc = sum(t * c for t, c in pmc) if all(t is not None for t, _ in pmc) else None
        out["roofline"] = {"bound": "hbm", "achieved": dom["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["gbs"] / HBM_PEAK_GBS,
                           "traffic": measured_traffic(dom), "kernel": "dwconv %s %s stage %d (N=%d)" % (dom["kernel"], dom["op"], dom["stage"], a.batch),
                           "avg_launch_ms": dom["ms"], "alg_bytes_per_launch": dom["alg_bytes"],
                           "valu_tflops_nominal": dom["gflop_nominal"] / dom["ms"],
                           # every launch within 10 % of the largest time per step (round 4's line named the stage-3 one-launch backward, 0.86)
                           "dominant_candidates": [{"kernel": "dwconv %s %s stage %d" % (k["kernel"], k["op"], k["stage"]), "step_ms": round(k["step_ms"], 4),
                                                    "frac": round(k["gbs"] / HBM_PEAK_GBS, 4)} for k in sorted(near, key=lambda k: -k["step_ms"])]}
        if out["roofline"]["traffic"]:                             # the dominant launch on the bytes it MOVES (PMC), beside `frac` on the per-op bytes of the ops it replaces
            out["roofline"]["frac_of_measured_traffic"] = out["roofline"]["traffic"] / dom["ms"] / 1e6 / HBM_PEAK_GBS
            out["roofline"]["note"] = ("frac prices the launch at SURVEY 8(d)'s per-op bytes of the ops it replaces; frac_of_measured_traffic is its measured HBM "
                                       "traffic (PMC, profiles/pmc_traffic.json) over the same time -- the bandwidth the kernel really draws")
        assert abs(hot_bytes - survey_8d_bytes(stages, a.batch, 4 if a.fp32_dwconv else 2)) <= 1e-6 * hot_bytes, "the launches timed do not add up to SURVEY 8(d)'s per-op bytes"
        # the PATH-level figures inside `roofline` as well (the north star's target is on the path): every dw-conv launch of a step, as the
        # step launches it (statistics-gathering forward variants included), algorithmic bytes / summed launch time
        out["roofline"].update({"path_frac": hot_bytes / hot_ms / 1e6 / HBM_PEAK_GBS, "path_achieved": hot_bytes / hot_ms / 1e6,
                                "path_ms_per_step": hot_ms, "path_alg_gb_per_step": hot_bytes / 1e9,
                                "path_lowest_launch_frac": min(k["gbs"] for k in kl) / HBM_PEAK_GBS,
                                "path_traffic_gb_per_step": (hot_bytes_pmc / 1e9) if hot_bytes_pmc else None})
        out["hot_path"] = {"dwconv_ms_per_step": hot_ms, "dwconv_alg_gb_per_step": hot_bytes / 1e9,
                           "dwconv_gbs": hot_bytes / hot_ms / 1e6, "dwconv_frac_of_hbm_peak": hot_bytes / hot_ms / 1e6 / HBM_PEAK_GBS,
                           "pricing": "SURVEY 8(d): 2*S*b per op and pass; a launch that replaces several ops is priced at the ops it replaces",
                           "dwconv_frac_incl_acc_reads": hot_bytes_acc / hot_ms / 1e6 / HBM_PEAK_GBS,
                           "dwconv_frac_of_measured_hbm_traffic": (hot_bytes_pmc / hot_ms / 1e6 / HBM_PEAK_GBS) if hot_bytes_pmc else None,
                           "lowest_kernel_frac": min(k["gbs"] for k in kl) / HBM_PEAK_GBS,
                           "share_of_step": hot_ms / ms_per_step,
                           "images_per_s_dwconv_only": a.batch / (hot_ms / 1e3),
                           "kernels": [{k2: (round(v, 4) if isinstance(v, float) else v) for k2, v in k.items()} for k in kl]}
    if rank == 0 and not a.no_mask_bench:
        out["mask_step"] = mask_step_bench(device, a.model, ks, a.only_L)
        if "roofline" in out:
            out["roofline"]["mask_apply_frac"] = out["mask_step"]["apply_frac_of_hbm_peak"]
            out["roofline"]["mask_update_frac"] = out["mask_step"]["update_frac_of_hbm_peak"]
    if rank == 0 and n_gpus == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(min(os.cpu_count() or 1, 32), stages, label)   # >32 threads only oversubscribes a 96-channel depthwise conv
    if rank == 0:
        try:                                                        # text that C libraries left in the C stdio buffer (RCCL's NCCL_DEBUG=VERSION banner) goes out FIRST,
            import ctypes                                           # so that the JSON line is the last line of stdout whatever the buffering
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
