#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mi in 3 2 1; do
  rm -rf /tmp/si && SLAK_SMALL_MIN_ITERS=$mi rocprofv3 --kernel-trace -d /tmp/si -o s -- python $R/tools/time_small.py > /tmp/si.log 2>&1
  echo "== SLAK_SMALL_MIN_ITERS=$mi"; python $R/tools/rocpd_summary.py $(find /tmp/si -name "*.db" | head -1) 10 | grep -E "slak" | cut -c1-60,110-175
done
