#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/team7; mkdir -p $O
timeout 120 python tools/time_team.py --check --small > $O/check.log 2>&1; echo "rc=$?" >> $O/check.log
timeout 120 python tools/time_team.py --check >> $O/check.log 2>&1; echo "rc=$?" >> $O/check.log
grep -c OK $O/check.log; grep -E "BAD|rc=" $O/check.log | cut -c1-250
if grep -q "rc=124" $O/check.log; then echo HANG; exit 1; fi
timeout 60 python tools/time_team.py 2>&1 | grep "tri " | tee $O/sweep.log
SLAK_STREAM_TRI=0 SLAK_TEAM_ALL=1 TEAM_SHAPES=1 timeout 60 python tools/time_team.py 2>&1 | grep "tri " | tee -a $O/sweep.log
