"""tools/time_all.py -- every dw-conv launch of the hot path AS THE MODEL RUNS IT (bench.hot_path_kernels: three-branch forward and
data-gradient launches, pair / three-branch weight-gradient launches), timed alone through the C ABI.  One line per launch.
    python tools/time_all.py [--model tiny|base] [--kernel 51] [--res 224] [--batch 128]
SLAK_TIME_ALL_REPS=n: n launches per kernel (PMC passes: tools/pmc_hot.sh)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="tiny"); ap.add_argument("--kernel", type=int, default=51); ap.add_argument("--res", type=int, default=224)
ap.add_argument("--batch", type=int, default=None)
a = ap.parse_args()
batch = a.batch or (64 if (a.model == "base" or a.res != 224) else 128)
dev = torch.device("cuda:0")
reps = int(os.environ.get("SLAK_TIME_ALL_REPS", "0")) or 30
stages = bench.stages_of(a.model, a.kernel, a.res)
kl = bench.hot_path_kernels(dev, batch, reps, torch.bfloat16, stages, plain_too=False)
tot = byt = 0.0
print("%-5s %-22s %-5s %-10s %9s %6s %8s %6s  %s" % ("stage", "kernel", "kind", "op", "us", "calls", "GB/s", "frac", "hip kernel"))
for k in kl:
    ms = k["ms"] * k["calls_per_step"]; tot += ms; byt += k["alg_bytes"] * k["calls_per_step"]
    gbs = k["alg_bytes"] / k["ms"] / 1e6
    print("%-5d %-22s %-5s %-10s %9.1f %6d %8.0f %6.3f  %s" % (k["stage"], k["kernel"], k["branch"], k["op"], k["ms"] * 1e3, k["calls_per_step"], gbs,
                                                       gbs / bench.HBM_PEAK_GBS, k.get("hip_kernel")))
if os.environ.get("SLAK_TIME_ALL_JSON"):
    import json
    keep = ("stage", "kernel", "branch", "op", "alg_bytes", "alg_bytes_incl_acc_read", "hip_kernel", "calls_per_step", "ms", "variant")
    json.dump([{k: e.get(k) for k in keep} for e in kl], open(os.environ["SLAK_TIME_ALL_JSON"], "w"), indent=1)
print("dw-conv hot path per step: %.3f ms, %.3f GB (SURVEY 8d), %.3f of the %.0f GB/s HBM peak" % (tot, byt / 1e9, byt / tot / 1e6 / bench.HBM_PEAK_GBS, bench.HBM_PEAK_GBS))
