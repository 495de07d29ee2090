#!/bin/bash
# round 6, GPU session 1: the N > 1 code path on one GPU -- gradients written into DDP's bucket views (A/B), SLaK-B, host profile, step table
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_distributed_gpu.py tests/test_model_reference_gpu.py tests/test_boundary.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="--steps 20 --warmup 5 --no-roofline --no-mask-bench --no-cpu-baseline"
run() { name=$1; shift; timeout 600 "$@" 2> $O/$name.err | grep '^{' | tail -1 > $O/$name.json; }
run cfg1 python bench.py $B
run cfg2_sparsity04 python bench.py $B --sparsity 0.4
run nccl_ws1 python bench.py $B --force-dist --sparsity 0.4
SLAK_GRAD_SLOTS=0 run nccl_ws1_noslots python bench.py $B --force-dist --sparsity 0.4
run nccl_ws1_refflags python bench.py $B --force-dist --sparsity 0.4 --ddp-reference-flags
run cfg3_slak_b python bench.py $B --model base
run nccl_ws1_slak_b python bench.py $B --model base --force-dist
SLAK_GRAD_SLOTS=0 run nccl_ws1_slak_b_noslots python bench.py $B --model base --force-dist
timeout 600 python bench.py $B --force-dist --sparsity 0.4 --host-profile > /dev/null 2> $O/host_profile_forcedist.txt
BENCH_ARGS="--no-mask-bench --no-roofline --force-dist --sparsity 0.4" timeout 600 bash tools/profile_bench.sh > /dev/null 2>&1
cp gpurun_out/sum/step_breakdown.txt $O/step_breakdown_forcedist.txt
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r06a")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f)); c = d["config"]
        print("%-32s %8.1f img/s %7.3f ms host %6.2f (in-region %6.2f) ddp=%s" % (os.path.basename(f), d["value"], d["ms_per_step"], c["host_enqueue_ms_per_step"], c["host_enqueue_ms_per_step_in_timed_region"], (c.get("ddp") or "")[-60:]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
head -24 $O/step_breakdown_forcedist.txt | cut -c1-200
grep -A 45 "cumulative" $O/host_profile_forcedist.txt | cut -c1-160 | head -60
