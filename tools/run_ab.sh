#!/bin/bash
# same-box A/B of the train step: default vs round-2 conv kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out/ab; mkdir -p $O
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-mask-bench --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default        ', round(d['ms_per_step'],3), round(d['value'],1))"
  SLAK_STREAM_TRI=0 SLAK_TEAM_TRI=0 python bench.py --no-cpu-baseline --no-mask-bench --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('round-2 kernels', round(d['ms_per_step'],3), round(d['value'],1))"
  SLAK_STREAM_TRI=0 python bench.py --no-cpu-baseline --no-mask-bench --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('team only      ', round(d['ms_per_step'],3), round(d['value'],1))"
done
timeout 300 python -m pytest tests/test_block_tail_gpu.py -m gpu -q --timeout 300 2>&1 | tail -3
