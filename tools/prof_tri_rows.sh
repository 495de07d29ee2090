#!/bin/bash
# needs a DEV build of the library (the shipped kernel compiles the experiments out): built here with SLAK_BUILD_DEFS="-DSLAK_TRIROWS_DEV -DSLAK_DEV_KNOBS"
export SLAK_BUILD_DEFS="-DSLAK_TRIROWS_DEV -DSLAK_DEV_KNOBS"; ( cd $GRAFT_REPO_ROOT && touch slak_amd/csrc/dwconv_mfma_tri_wgrad_rows.hip && python -m slak_amd.build > /dev/null 2>&1 )
# dev: true kernel durations (rocprofv3 --kernel-trace) of the one-launch weight gradient under SLAK_TRIROWS_DBG settings
#   usage: tools/prof_tri_rows.sh "N C H K" dbg...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; SHAPE=$1; shift
for st in "$@"; do
  rm -rf /tmp/kv && env WITH_PAIR=1 SLAK_TRIROWS_DBG=$st rocprofv3 --kernel-trace -d /tmp/kv -o kv -- python $R/tools/run_tri_rows.py $SHAPE > /tmp/kv.log 2>&1
  echo "== SLAK_TRIROWS_DBG=$st  shape $SHAPE"
  python $R/tools/rocpd_summary.py $(find /tmp/kv -name "*.db" | head -1) --top 12 | grep -E "wgrad" | cut -c1-64,110-170
done
