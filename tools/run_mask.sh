#!/bin/bash
# mask-step tests first (fail fast), then the whole GPU suite, then the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/mask; mkdir -p $O
timeout 600 python -m pytest tests/test_masking_gpu.py -q --timeout 300 -x > $O/pytest_mask.log 2>&1; tail -15 $O/pytest_mask.log
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x --deselect tests/test_masking_gpu.py > $O/pytest.log 2>&1; tail -8 $O/pytest.log
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; cut -c1-600 $O/bench.json; tail -3 $O/bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/mask/bench.json').read())
print(json.dumps(d.get('mask_step'))[:1500])
print(json.dumps(d.get('hot_path'))[:600])
P
