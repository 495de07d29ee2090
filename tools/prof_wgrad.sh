#!/bin/bash
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 0 1 2; do
  rm -rf /tmp/pw$d
  SLAK_LW_DBG=$d rocprofv3 --kernel-trace --stats -d /tmp/pw$d -o t -- python $R/tools/time_wgrad_gemm.py > /dev/null 2>&1
  echo "== SLAK_LW_DBG=$d"
  DB=$(find /tmp/pw$d -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB --top 40 | grep -E "linear_wgrad" | cut -c1-60,100-200
done
