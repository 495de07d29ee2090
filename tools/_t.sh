cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_fused_launches_gpu.py tests/test_mfma_gpu.py tests/test_dispatch_gpu.py tests/test_block_tail_gpu.py -q --timeout 600 -x 2>&1 | tail -3
timeout 300 python tools/time_all.py 2>&1 | grep -v amdgpu
timeout 600 python bench.py --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 | cut -c1-200
