#!/bin/bash
# round 4, call 1: changed tests, valid step breakdown (markers), host profile
cd $GRAFT_REPO_ROOT; O=gpurun_out/c1; mkdir -p $O
timeout 900 python -m pytest tests/test_fp32_mfma_gpu.py tests/test_fused_launches_gpu.py tests/test_masking_gpu.py tests/test_boundary.py -q -m gpu --timeout 600 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
BENCH_ARGS="--no-mask-bench" timeout 600 bash tools/profile_bench.sh > $O/profile_bench.log 2>&1
cp gpurun_out/sum/bench_kernel_stats.txt gpurun_out/sum/step_breakdown.txt gpurun_out/sum/bench_under_rocprof.json $O/ 2>/dev/null
head -30 $O/step_breakdown.txt
timeout 600 python bench.py --host-profile --no-cpu-baseline --no-mask-bench --no-roofline > $O/bench_host.json 2> $O/host_profile.txt
cut -c1-400 $O/bench_host.json
