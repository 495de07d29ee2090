#!/usr/bin/env python3
"""Steady-state per-step kernel breakdown from a rocprofv3 --kernel-trace database of bench.py (last 4 steps,
delimited by the fused-AdamW launches).   python tools/step_breakdown.py gpurun_out/prof_step/step_results.db [--top 40]"""
import collections, sqlite3, sys
db = sys.argv[1]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
c = sqlite3.connect(db)
rows = list(c.execute("select name, start, end from kernels order by start"))
adam = [r for r in rows if "multi_tensor_apply" in r[0] or "slak::adamw_kernel" in r[0]]
steps = []
for r in adam:
    if not steps or r[1] - steps[-1][-1] > 10e6: steps.append([r[1]])
    else: steps[-1].append(r[1])
nst = 4
t0, t1 = steps[-1 - nst][-1], steps[-1][-1]
agg = collections.defaultdict(lambda: [0, 0])
for n, s, e in rows:
    if t0 < s <= t1 + 1e5:
        agg[n][0] += 1; agg[n][1] += e - s
tot = sum(v[1] for v in agg.values())
def cat(n):
    if "slak::adamw" in n or "slak::ema" in n: return "optimizer"
    if "slak::dwconv" in n or "toeplitz" in n: return "slak dwconv"
    if "slak::ln_" in n or "slak::scale_res" in n or "block_tail" in n: return "slak block tail"
    if "slak::linear_" in n or "slak::gelu_" in n: return "slak pointwise (linear_nt, linear_wgrad, gelu_bwd)"
    if "slak::bn3" in n: return "slak branch BatchNorm (bn3)"
    if "slak::mask" in n: return "slak mask step"
    if "slak::" in n: return "slak other"
    if n.startswith("Cijk"): return "hipblaslt gemm"
    if "BatchNorm" in n or "batch_norm" in n: return "batchnorm"
    if "layer_norm" in n or "LayerNorm" in n or "GammaBeta" in n or "cuComputeGradInput" in n: return "layernorm (torch)"
    if "conv" in n.lower() or "Im2d2Col" in n or "Col2Im" in n or "transpose" in n.lower() or "igemm" in n.lower(): return "MIOpen conv (stem/downsample)"
    if "elementwise" in n or "vectorized" in n: return "elementwise"
    if "reduce" in n: return "reduce"
    if "multi_tensor" in n: return "optimizer"
    return "other"
cats = collections.defaultdict(float)
for n, v in agg.items(): cats[cat(n)] += v[1]
print("# %s: window %.2f ms/step, GPU busy %.2f ms/step (last %d steps)" % (db, (t1 - t0) / 1e6 / nst, tot / 1e6 / nst, nst))
for k, v in sorted(cats.items(), key=lambda x: -x[1]): print("%-32s %7.2f ms/step" % (k, v / 1e6 / nst))
print()
for n, v in sorted(agg.items(), key=lambda x: -x[1][1])[:top]:
    print("%-130s %6.1f/step %8.3f ms/step" % (n[:130], v[0] / nst, v[1] / 1e6 / nst))
