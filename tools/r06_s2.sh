#!/bin/bash
# round 6, GPU session 2: where the N > 1 path's remaining cost sits (SyncBN exchange variants, DDP alone), SLaK-B host profile
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_distributed_gpu.py tests/test_round6_entry_points_gpu.py tests/test_model_reference_gpu.py tests/test_block_tail_gpu.py tests/test_linear_skinny_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
B="--steps 20 --warmup 5 --no-roofline --no-mask-bench --no-cpu-baseline"
run() { name=$1; shift; timeout 600 "$@" 2> $O/$name.err | grep '^{' | tail -1 > $O/$name.json; }
run cfg2_sparsity04 python bench.py $B --sparsity 0.4
run nccl_ws1 python bench.py $B --force-dist --sparsity 0.4
SLAK_BN_BWD_ASYNC=0 run nccl_ws1_bwd_sync python bench.py $B --force-dist --sparsity 0.4
SLAK_FORCE_BN_EXCHANGE=0 run nccl_ws1_no_exchange python bench.py $B --force-dist --sparsity 0.4
run cfg3_slak_b python bench.py $B --model base
run nccl_ws1_slak_b python bench.py $B --model base --force-dist
SLAK_BN_BWD_ASYNC=0 run nccl_ws1_slak_b_bwd_sync python bench.py $B --model base --force-dist
SLAK_FORCE_BN_EXCHANGE=0 run nccl_ws1_slak_b_no_exchange python bench.py $B --model base --force-dist
timeout 600 python bench.py $B --model base --force-dist --host-profile > /dev/null 2> $O/host_profile_slak_b_forcedist.txt
timeout 600 python bench.py $B --model base --host-profile > /dev/null 2> $O/host_profile_slak_b.txt
python - <<'PY'
import json, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r06b")
for f in sorted(glob.glob(O + "/*.json")):
    try:
        d = json.load(open(f)); c = d["config"]
        print("%-36s %8.1f img/s %7.3f ms host %6.2f (in-region %6.2f)" % (os.path.basename(f), d["value"], d["ms_per_step"], c["host_enqueue_ms_per_step"], c["host_enqueue_ms_per_step_in_timed_region"]))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
for f in host_profile_slak_b_forcedist host_profile_slak_b; do echo "== $f"; grep -A 40 "cumulative" $O/$f.txt | cut -c1-150 | head -48; done
