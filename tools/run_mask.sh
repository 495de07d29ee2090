#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/mask; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_masking_gpu.py -q --timeout 300 -x 2>&1 | tail -3
timeout 300 python tools/time_mask.py
UNREACHABLE=1 timeout 300 python tools/time_mask.py
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o mask -- python tools/time_mask.py > $O/prof.log 2>&1
python - <<'P'
import csv,glob
for f in glob.glob('gpurun_out/mask/prof/**/*kernel_stats.csv', recursive=True):
    for i,r in enumerate(csv.DictReader(open(f))):
        if 'mask' in r['Name']: print(r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
P
