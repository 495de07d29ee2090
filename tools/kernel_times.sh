#!/bin/bash
# Dev: true per-kernel durations (rocprofv3 kernel trace) of the conv microbenchmark; the event-timed loop of tools/time_all.py is
# host-bound below ~14 us per call.   usage: tools/kernel_times.sh [out.txt]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt && rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/time_all.py > /tmp/kt.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) 40 | grep -E "slak|^#|^kernel" | cut -c1-200
