#!/bin/bash
# HBM traffic of every kernel of the train step (tools/pmc_step.py): two PMC passes + one plain kernel trace of bench.py --markers  -> gpurun_out/sum/pmc_step.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/sum
A="--markers --steps 3 --warmup 2 --prime 3 --no-cpu-baseline --no-roofline --no-mask-bench $BENCH_ARGS"
rm -rf /tmp/ps_f /tmp/ps_w /tmp/ps_t
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/ps_f -o p -- python $R/bench.py $A > /tmp/ps_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/ps_w -o p -- python $R/bench.py $A > /tmp/ps_w.log 2>&1
rocprofv3 --kernel-trace -d /tmp/ps_t -o p -- python $R/bench.py $A > /tmp/ps_t.log 2>&1
python $R/tools/pmc_step.py $(find /tmp/ps_f -name "*.db" | head -1) $(find /tmp/ps_w -name "*.db" | head -1) $(find /tmp/ps_t -name "*.db" | head -1) --steps 3 --top 60 > $R/gpurun_out/sum/pmc_step.txt 2>&1
head -50 $R/gpurun_out/sum/pmc_step.txt | cut -c1-170
