#!/bin/bash
# round 6, GPU session 3: the N > 1 path after the round's changes (gradients into the bucket views, ncclAvg hook, exchange on the compute stream, fewer launches)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
timeout 1700 python -m pytest tests/test_distributed_gpu.py tests/test_round6_entry_points_gpu.py tests/test_model_reference_gpu.py tests/test_block_tail_gpu.py tests/test_linear_skinny_gpu.py tests/test_reference_engine_gpu.py -q -m gpu > $O/pytest.log 2>&1; tail -6 $O/pytest.log
B="--steps 20 --warmup 5 --no-roofline --no-mask-bench --no-cpu-baseline"
run() { name=$1; shift; timeout 600 "$@" 2> $O/$name.err | grep '^{' | tail -1 > $O/$name.json; }
run cfg2_sparsity04 python bench.py $B --sparsity 0.4
run nccl_ws1 python bench.py $B --force-dist --sparsity 0.4
run nccl_ws1_refflags python bench.py $B --force-dist --sparsity 0.4 --ddp-reference-flags
SLAK_BN_BWD_ASYNC=1 run nccl_ws1_bwd_async python bench.py $B --force-dist --sparsity 0.4
SLAK_GRAD_SLOTS=0 run nccl_ws1_noslots python bench.py $B --force-dist --sparsity 0.4
run cfg3_slak_b python bench.py $B --model base
run nccl_ws1_slak_b python bench.py $B --model base --force-dist
run nccl_ws1_slak_b_refflags python bench.py $B --model base --force-dist --ddp-reference-flags
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r06c")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f)); c = d["config"]
        print("%-36s %8.1f img/s %7.3f ms host %6.2f (in-region %6.2f) %s" % (os.path.basename(f), d["value"], d["ms_per_step"], c["host_enqueue_ms_per_step"], c["host_enqueue_ms_per_step_in_timed_region"], (c.get("ddp") or "")[-70:]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
grep -h "Error\|error" $O/*.err | grep -v amdgpu | head
