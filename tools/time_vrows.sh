#!/bin/bash
# dev: kernel durations of the stage-1 vertical weight gradient (tools/one_shape.py) under rocprofv3 for SLAK_VROWS_DBG values
# (1 = no k-loop, 2 = no DMA, 4 = no epilogue; N = 6 is the min column, N = 128 the max)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for st in ${@:-0}; do
  rm -rf /tmp/kv && SLAK_VROWS_DBG=$st rocprofv3 --kernel-trace -d /tmp/kv -o kv -- python $R/tools/one_shape.py > /tmp/kv.log 2>&1
  echo "== SLAK_VROWS_DBG=$st"
  python $R/tools/rocpd_summary.py $(find /tmp/kv -name "*.db" | head -1) --top 12 | grep -E "vrows" | cut -c1-70,110-190
done
