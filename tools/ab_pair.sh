#!/bin/bash
# dev: ablations of the stage-1 pair weight gradient (SLAK_VROWS_DBG bits: 1 no k-loop, 2 no DMA, 4 no epilogue, 8 no barrier/wait)
cd $GRAFT_REPO_ROOT
for v in "" "SLAK_VROWS_DBG=2" "SLAK_VROWS_DBG=10" "SLAK_VROWS_DBG=14" "SLAK_VROWS_DBG=1" "SLAK_VROWS_DBG=3" "SLAK_VROWS_DBG=7" "SLAK_VROWS_DBG=4" \
         "SLAK_VROWS_WGS=1" "SLAK_VROWS_DBG=2 SLAK_VROWS_WGS=1" "SLAK_VROWS_DBG=10 SLAK_VROWS_WGS=1" "SLAK_VROWS_DBG=14 SLAK_VROWS_WGS=1" "SLAK_VROWS_NB=3" ; do
  env $v python tools/ab_pair.py
done
