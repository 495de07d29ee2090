#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/host; mkdir -p $O
timeout 900 python -m pytest tests/test_block_tail_gpu.py tests/test_model_reference_gpu.py tests/test_masking_gpu.py tests/test_optim_ema_gpu.py tests/test_distributed_gpu.py -q --timeout 600 -x 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-mask-bench --no-roofline 2> /dev/null | tail -1 > $O/dense.json
timeout 600 python bench.py --sparsity 0.4 --no-cpu-baseline --no-mask-bench --no-roofline 2> /dev/null | tail -1 > $O/sparse.json
python - <<'P'
import json
for n in ("dense","sparse"):
    d=json.loads(open(f"gpurun_out/host/{n}.json").read()); print(n, d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"])
P
