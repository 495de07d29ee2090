#!/bin/bash
# HBM traffic of every dw-conv launch of the hot path from PMC counters (two separate --pmc passes) -> gpurun_out/sum/pmc_traffic.{txt,json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sum
rm -rf /tmp/pmc_fetch /tmp/pmc_write
SLAK_TIME_ALL_REPS=3 SLAK_TIME_ALL_JSON=/tmp/pmc_entries.json rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_fetch -o pmc -- python $R/tools/time_all.py > /tmp/pmc_f.log 2>&1
SLAK_TIME_ALL_REPS=3 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_write -o pmc -- python $R/tools/time_all.py > /tmp/pmc_w.log 2>&1
cd $R && python tools/pmc_traffic.py $(find /tmp/pmc_fetch -name "*.db" | head -1) $(find /tmp/pmc_write -name "*.db" | head -1) /tmp/pmc_entries.json > gpurun_out/sum/pmc_traffic.txt 2>&1
cp profiles/pmc_traffic.json gpurun_out/sum/pmc_traffic.json
tail -40 gpurun_out/sum/pmc_traffic.txt
