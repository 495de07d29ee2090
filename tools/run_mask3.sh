#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/mask; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_masking_gpu.py tests/test_fused_launches_gpu.py::test_shapes_without_a_three_branch_launch_say_so -q --timeout 300 -x 2>&1 | tail -3
timeout 300 python tools/time_mask.py
UNREACHABLE=1 timeout 300 python tools/time_mask.py
timeout 600 python bench.py --sparsity 0.4 --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 > $O/bench_cfg2.json; cut -c1-330 $O/bench_cfg2.json
timeout 600 python bench.py --sparsity 0.4 --update-frequency 1 --steps 10 --warmup 3 --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 > $O/bench_cfg2_u1.json; cut -c1-330 $O/bench_cfg2_u1.json
