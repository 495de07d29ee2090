#!/bin/bash
# dev: where the time of dwconv_mfma_wide_tri goes (library built with -DSLAK_WT_DEV -DSLAK_DEV_KNOBS): SLAK_WT_DBG 1 no MFMA tiles, 2 no transposes, 4 no DMA, 8 no stores
cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 8 6 14 15 7; do
  echo "SLAK_WT_DBG=$d: $(SLAK_WT_DBG=$d python tools/time_all.py --kernel 61 --res 384 2>/dev/null | grep -E '^1 .*tri ' | awk '{print $4, $5, $6}' | tr '\n' ' ')"
done
