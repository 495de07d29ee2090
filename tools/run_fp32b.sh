#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/fp32; mkdir -p $O
timeout 900 python -m pytest tests/test_fp32_mfma_gpu.py tests/test_mfma_gpu.py::test_mfma_is_what_auto_runs_for_lowp -q --timeout 300 -x -s 2>&1 | tail -5
timeout 600 python tools/time_fp32.py 2>&1 | head -3
