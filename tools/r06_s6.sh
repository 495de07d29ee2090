#!/bin/bash
# round 6, GPU session 6: the three-branch kernel for 64 < H, W <= 96 (configs[4] stage 1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_fused_launches_gpu.py tests/test_dispatch_gpu.py tests/test_distributed_gpu.py -x -q -m gpu -k "wide or dispatch or say_so or bench_shapes or ddp or two_ranks or bench_n or force_dist" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for i in 1 2; do
  echo "WIDE_TRI=1"; python tools/time_all.py --kernel 61 --res 384 2>/dev/null | grep -E "^1 |hot path"
  echo "WIDE_TRI=0"; SLAK_WIDE_TRI=0 python tools/time_all.py --kernel 61 --res 384 2>/dev/null | grep -E "^1 |hot path"
done
B="--steps 20 --warmup 5 --no-mask-bench --no-cpu-baseline --no-roofline"
run() { name=$1; shift; timeout 600 "$@" 2> $O/$name.err | grep '^{' | tail -1 > $O/$name.json; }
run cfg4 python bench.py $B --kernel 61 --res 384
SLAK_WIDE_TRI=0 run cfg4_wide_tri_off python bench.py $B --kernel 61 --res 384
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r06f")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f)); c = d["config"]
        print("%-36s %8.1f img/s %7.3f ms host %6.2f" % (os.path.basename(f), d["value"], d["ms_per_step"], c["host_enqueue_ms_per_step"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
