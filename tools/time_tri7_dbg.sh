#!/bin/bash
# dev: phases of the 7x7 quad weight-gradient kernel (SLAK_QW_DBG: 1 no compute, 2 no DMA, 4 no diagonal sums, 8 no octets)
for d in 0 1 2 3 4 7 8 12; do
  echo -n "dbg=$d  "; SLAK_QW_DBG=$d python tools/time_tri7.py | grep wgrad
done
