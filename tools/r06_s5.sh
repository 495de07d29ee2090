#!/bin/bash
# round 6, GPU session 5: tri_wgrad_rows SD A/B on one box; the N > 1 path with the all-reduce issued from C++
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
for i in 1 2 3; do
  echo "SD=1"; python tools/time_all.py 2>/dev/null | grep -E "^1 |hot path"
  echo "SD=0"; SLAK_TRI_ROWS_SD=0 python tools/time_all.py 2>/dev/null | grep -E "^1 |hot path"
done
timeout 900 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -3
B="--steps 20 --warmup 5 --no-roofline --no-mask-bench --no-cpu-baseline"
run() { name=$1; shift; timeout 600 "$@" 2> $O/$name.err | grep '^{' | tail -1 > $O/$name.json; }
run cfg2_sparsity04 python bench.py $B --sparsity 0.4
run nccl_ws1 python bench.py $B --force-dist --sparsity 0.4
SLAK_PG_DIRECT=0 run nccl_ws1_pg_py python bench.py $B --force-dist --sparsity 0.4
SLAK_TRI_ROWS_SD=0 run cfg2_sparsity04_sd0 python bench.py $B --sparsity 0.4
run cfg3_slak_b python bench.py $B --model base
run nccl_ws1_slak_b python bench.py $B --model base --force-dist
SLAK_PG_DIRECT=0 run nccl_ws1_slak_b_pg_py python bench.py $B --model base --force-dist
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r06e")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f)); c = d["config"]
        print("%-36s %8.1f img/s %7.3f ms host %6.2f (in-region %6.2f)" % (os.path.basename(f), d["value"], d["ms_per_step"], c["host_enqueue_ms_per_step"], c["host_enqueue_ms_per_step_in_timed_region"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
