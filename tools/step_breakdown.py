#!/usr/bin/env python3
"""Per-step kernel breakdown of EXACTLY the K timed steps of bench.py from a rocprofv3 --kernel-trace database.

    rocprofv3 --kernel-trace -d /tmp/pb -o bench -- python bench.py --markers --steps K ...
    python tools/step_breakdown.py <results.db> --steps K [--top 130] [--expect NAME=CALLS_PER_STEP ...]

bench.py --markers launches `slak::marker_kernel` with grid 2 (id 1) right before the first timed step and with grid 3 (id 2) right behind
the last one; this tool takes the dispatches that START between the two -- no warm-up steps, no GEMM burn of the roofline section, no
micro-benchmarks, no mask-step measurement.  Sanity: the calls per step of a few kernels whose count is known from the model (SLaK-T: 18
blocks -> 18 `bn3_apply_fwd`, 18 `bn3_apply_bwd`, one `adamw_kernel`) are asserted (`--expect`; defaults for SLaK-T / -S / -B by the block
count found), and every number DESIGN.md quotes from this table is a line of its output."""
import collections
import sqlite3
import sys


def arg(name, default=None, cast=str):
    return cast(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


def cat(n):
    if "marker_kernel" in n: return "marker"
    if "slak::adamw" in n or "slak::ema" in n: return "optimizer (slak adamw/ema)"
    if "slak::dwconv" in n or "toeplitz" in n: return "slak dwconv (the hot path)"
    if "slak::ln_" in n or "slak::scale_res" in n or "block_tail" in n or "slak::stem" in n or "channel_sums_kernel" in n or "fill_channel_bias_kernel" in n: return "slak block tail (LN/permute, scale+residual, patchify)"
    if "slak::linear_" in n or "slak::gelu_" in n: return "slak pointwise (linear_nt, linear_wgrad, gelu_bwd)"
    if "slak::bn3" in n: return "slak branch BatchNorm (bn3)"
    if "slak::mask" in n: return "slak mask step"
    if "slak::" in n: return "slak other"
    if n.startswith("Cijk") or n.startswith("Custom_Cijk"): return "hipblaslt gemm"
    if "multi_tensor" in n: return "torch multi_tensor_apply (foreach ops)"
    if "BatchNorm" in n or "batch_norm" in n: return "batchnorm (torch)"
    if "layer_norm" in n or "LayerNorm" in n or "GammaBeta" in n or "cuComputeGradInput" in n: return "layernorm (torch)"
    if "conv" in n.lower() or "Im2d2Col" in n or "Col2Im" in n or "transpose" in n.lower() or "igemm" in n.lower(): return "MIOpen conv"
    if "copyBuffer" in n or "fillBuffer" in n: return "runtime copy/fill"
    if "elementwise" in n or "vectorized" in n: return "torch elementwise"
    if "reduce" in n: return "torch reduce"
    if "softmax" in n.lower() or "nll_loss" in n.lower(): return "loss (torch)"
    return "other"


def main():
    db = sys.argv[1]
    steps = arg("--steps", None, int)
    top = arg("--top", 130, int)
    if steps is None:
        sys.exit("--steps K (the K of the profiled bench.py run) is required")
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start"))
    # rocpd's grid_x is in work-items on some versions and in workgroups on others: normalise with the workgroup size (64)
    def marker_id(r):
        g, w = r[3], r[4] or 64
        n = g // w if (g % w == 0 and g >= w and g // w in (2, 3)) else g
        return n - 1
    marks = [(marker_id(r), r) for r in rows if "marker_kernel" in r[0]]
    m1 = [r for i, r in marks if i == 1]
    m2 = [r for i, r in marks if i == 2]
    if len(m1) != 1 or len(m2) != 1:
        sys.exit("expected exactly one marker 1 and one marker 2 in the trace (bench.py --markers), found %d / %d" % (len(m1), len(m2)))
    t0, t1 = m1[0][2], m2[0][1]
    agg = collections.defaultdict(lambda: [0, 0])
    first = last = None
    for n, s, e, _, _ in rows:
        if t0 <= s < t1 and "marker_kernel" not in n:
            agg[n][0] += 1; agg[n][1] += e - s
            first = s if first is None else first
            last = e if last is None or e > last else last
    tot = sum(v[1] for v in agg.values())

    def calls(sub):
        return sum(v[0] for n, v in agg.items() if sub in n) / steps

    blocks = calls("bn3_apply_fwd")
    expect = {"bn3_apply_fwd": blocks, "bn3_apply_bwd": blocks, "slak::adamw_kernel": 1.0}
    if blocks not in (18.0, 36.0):
        sys.exit("bn3_apply_fwd runs %.3f times per step: not a whole model forward per step (18 blocks SLaK-T, 36 SLaK-S/-B) -- wrong window?" % blocks)
    for kv in [a for i, a in enumerate(sys.argv) if i > 0 and sys.argv[i - 1] == "--expect"]:
        k, v = kv.split("="); expect[k] = float(v)
    for k, v in expect.items():
        got = calls(k)
        if k == "slak::adamw_kernel" and got == 0:
            continue                                             # --torch-adamw runs
        assert abs(got - v) < 1e-9, "%s: %.3f calls per step in the window, expected %.3f" % (k, got, v)
    print("# %s" % db)
    print("# window: marker 1 -> marker 2 = the %d timed steps of bench.py --markers; %.3f ms per step wall (first dispatch start to last dispatch end), "
          "GPU busy (sum of kernel durations) %.3f ms per step, %d dispatches per step, %d blocks" % (steps, (last - first) / 1e6 / steps, tot / 1e6 / steps,
                                                                                                  sum(v[0] for v in agg.values()) / steps, int(blocks)))
    cats = collections.defaultdict(lambda: [0, 0.0])
    for n, v in agg.items():
        cats[cat(n)][0] += v[0]; cats[cat(n)][1] += v[1]
    print("%-62s %10s %12s %7s" % ("subsystem", "calls/step", "ms/step", "share"))
    for k, v in sorted(cats.items(), key=lambda x: -x[1][1]):
        print("%-62s %10.1f %12.3f %6.1f%%" % (k, v[0] / steps, v[1] / 1e6 / steps, 100.0 * v[1] / tot))
    print()
    print("%-150s %10s %10s %10s" % ("kernel", "calls/step", "avg us", "ms/step"))
    for n, v in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
        print("%-150s %10.2f %10.2f %10.3f" % (n[:150], v[0] / steps, v[1] / 1e3 / v[0], v[1] / 1e6 / steps))


if __name__ == "__main__":
    main()
