#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/team6; mkdir -p $O
timeout 120 python tools/time_team.py --check --small > $O/check.log 2>&1; echo "rc=$?" >> $O/check.log
timeout 120 python tools/time_team.py --check >> $O/check.log 2>&1; echo "rc=$?" >> $O/check.log
grep -c OK $O/check.log; grep -E "BAD|rc=" $O/check.log | cut -c1-250
if grep -q "rc=124" $O/check.log; then echo HANG; exit 1; fi
timeout 60 python tools/time_team.py 2>&1 | grep "tri " | tee $O/sweep.log
export TEAM_SHAPES=1
for nb in 2 3 4; do SLAK_TEAM_NB=$nb timeout 60 python tools/time_team.py 2>&1 | grep "tri "; done | tee -a $O/sweep.log
for d in 1 2 4 6; do SLAK_TEAM_DBG=$d timeout 60 python tools/time_team.py 2>&1 | grep "tri "; done | tee -a $O/sweep.log
SLAK_TEAM_DBG=16 timeout 100 python tools/team_timeline.py 2>&1 | tail -12
