#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/host; mkdir -p $O
timeout 600 python bench.py --no-cpu-baseline --no-mask-bench --no-roofline --host-profile 2> $O/dense.prof | tail -1 > $O/dense.json
timeout 600 python bench.py --sparsity 0.4 --no-cpu-baseline --no-mask-bench --no-roofline --host-profile 2> $O/sparse.prof | tail -1 > $O/sparse.json
python - <<'P'
import json
for n in ("dense","sparse"):
    d=json.loads(open(f"gpurun_out/host/{n}.json").read()); print(n, d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"])
P
