#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/fp32; mkdir -p $O
timeout 900 python -m pytest tests/test_fp32_mfma_gpu.py -q --timeout 300 -x -s > $O/pytest_fp32.log 2>&1; tail -25 $O/pytest_fp32.log
timeout 900 python -m pytest tests/test_dwconv_gpu.py tests/test_mfma_gpu.py -q --timeout 600 -x > $O/pytest_reg.log 2>&1; tail -5 $O/pytest_reg.log
timeout 600 python tools/time_fp32.py > $O/time_fp32.txt 2>&1; cat $O/time_fp32.txt
