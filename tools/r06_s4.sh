#!/bin/bash
# round 6, GPU session 4: idle time inside the step, N = 1 against the N > 1 code path (rocprofv3 traces), + first numbers of the tri_wgrad_rows SD variant
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_fused_launches_gpu.py tests/test_dispatch_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for v in n1 dist; do
  A="--sparsity 0.4"; [ $v = dist ] && A="--force-dist --sparsity 0.4"
  rm -rf /tmp/pb_$v && rocprofv3 --kernel-trace -d /tmp/pb_$v -o bench -- python $R/bench.py --markers --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-mask-bench $A > /tmp/pb_$v.log 2>&1
  DB=$(find /tmp/pb_$v -name "*.db" | head -1)
  python $R/tools/step_gaps.py $DB --steps 6 --top 30 > $O/step_gaps_$v.txt
  python $R/tools/step_breakdown.py $DB --steps 6 --top 40 > $O/step_breakdown_$v.txt
done
cd $R
head -34 $O/step_gaps_n1.txt | cut -c1-170; head -34 $O/step_gaps_dist.txt | cut -c1-170
python tools/time_all.py 2>/dev/null | tail -30
SLAK_TRI_ROWS_SD=0 python tools/time_all.py 2>/dev/null | grep -i "stage 1\|56" | tail -8
