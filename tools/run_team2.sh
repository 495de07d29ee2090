#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/team2; mkdir -p $O
timeout 120 python tools/time_team.py --check --small > $O/check.log 2>&1; timeout 120 python tools/time_team.py --check >> $O/check.log 2>&1
grep -c OK $O/check.log; grep BAD $O/check.log
export TEAM_SHAPES=2
for dual in 0 1; do for sg in 0 2 4 6 8 12; do
  SLAK_TEAM_DUAL=$dual SLAK_TEAM_STAGGER=$sg timeout 60 python tools/time_team.py 2>&1 | grep "tri "
done; done | tee $O/sweep.log
for d in 1 2; do SLAK_TEAM_DBG=$d SLAK_TEAM_STAGGER=4 timeout 60 python tools/time_team.py 2>&1 | grep "tri "; done | tee -a $O/sweep.log
