#!/bin/bash
# rocprofv3 kernel trace of the bench -> gpurun_out/sum/{bench_kernel_stats.txt, step_breakdown.txt, bench_under_rocprof.json}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sum
rm -rf /tmp/pb && rocprofv3 --kernel-trace --stats -d /tmp/pb -o bench -- python $R/bench.py --markers --steps 6 --warmup 2 --no-cpu-baseline --kernel-reps 3 $BENCH_ARGS > /tmp/pb.log 2>&1
grep '^{' /tmp/pb.log | tail -1 > $R/gpurun_out/sum/bench_under_rocprof.json
DB=$(find /tmp/pb -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB --top 70 > $R/gpurun_out/sum/bench_kernel_stats.txt
python $R/tools/step_breakdown.py $DB --steps 6 --top 130 > $R/gpurun_out/sum/step_breakdown.txt
head -16 $R/gpurun_out/sum/step_breakdown.txt
