#!/bin/bash
# Hardware-timestamp durations (rocprofv3 --kernel-trace) of every dw-conv launch of the hot path as the step launches it.
#   usage: tools/kernel_times.sh out.txt [time_all args]      (event-timed Python loops are host-bound below ~14 us per call)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$1; shift
rm -rf /tmp/kt && SLAK_TIME_ALL_JSON=/tmp/kt_entries.json rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/tools/time_all.py "$@" > /tmp/kt.log 2>&1
python $R/tools/kernel_times.py $(find /tmp/kt -name "*.db" | head -1) /tmp/kt_entries.json >> $OUT 2>&1
