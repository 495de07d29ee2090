#!/bin/bash
# needs a DEV build of the library (the shipped kernel compiles the experiments out): built here with SLAK_BUILD_DEFS="-DSLAK_TRIROWS_DEV -DSLAK_DEV_KNOBS"
export SLAK_BUILD_DEFS="-DSLAK_TRIROWS_DEV -DSLAK_DEV_KNOBS"; ( cd $GRAFT_REPO_ROOT && touch slak_amd/csrc/dwconv_mfma_tri_wgrad_rows.hip && python -m slak_amd.build > /dev/null 2>&1 )
# The shader clock inside the one-launch weight gradient (128 x 96 x 56 x 56, bf16) as a function of how many CUs work and whether they stream from
# HBM: s_memtime cycles / s_memrealtime (100 MHz) of workgroup 0.  36 = the kernel without its epilogue; 38 = without DMA as well; 100 = DMA from
# one cached MiB.  -> profiles/rNN_power_clock_probe.txt
cd $GRAFT_REPO_ROOT
for w in 256 128 64 32; do for d in 36 100 38; do printf "workgroups %3d  SLAK_TRIROWS_DBG=%-3d  " $w $d; SLAK_TRIROWS_WGS=$w SLAK_TRIROWS_DBG=$d python tools/clk_tri_rows.py 2>&1 | grep "wg 0"; done; done
