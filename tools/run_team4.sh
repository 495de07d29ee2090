#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/team4; mkdir -p $O
SLAK_TEAM_DUAL=0 timeout 120 python tools/time_team.py --check > $O/check.log 2>&1; echo "rc=$?" >> $O/check.log
grep -c OK $O/check.log; grep -E "BAD|rc=" $O/check.log | cut -c1-200
if grep -q "rc=124" $O/check.log; then echo HANG; exit 1; fi
export TEAM_SHAPES=2
for dual in 0 1; do SLAK_TEAM_DUAL=$dual timeout 60 python tools/time_team.py 2>&1 | grep "tri "; done | tee $O/sweep.log
for d in 1 2 6; do SLAK_TEAM_DBG=$d timeout 60 python tools/time_team.py 2>&1 | grep "tri "; done | tee -a $O/sweep.log
