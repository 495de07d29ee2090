#!/bin/bash
# per-kernel durations of the block-tail kernels (tools/time_tail.py) with the register-tile kernels on and off
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in 1 0; do
  rm -rf /tmp/pt$m
  SLAK_TAIL_REG=$m rocprofv3 --kernel-trace --stats -d /tmp/pt$m -o t -- python $R/tools/time_tail.py > /dev/null 2>&1
  echo "== SLAK_TAIL_REG=$m"
  DB=$(find /tmp/pt$m -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $DB --top 40 | grep -E "ln_nchw|scale_residual|block_tail_reduce|Name|name"
done
