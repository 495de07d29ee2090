#!/bin/bash
cd $GRAFT_REPO_ROOT; export TEAM_SHAPES=1
for d in 0 2 4 8 32 40 64 72 106 110; do SLAK_TEAM_DBG=$d timeout 60 python tools/time_team.py 2>&1 | grep "tri fwd"; done
