#!/bin/bash
# first contact with the team kernels: parity on small + bench shapes under a timeout, then timings and ablations
cd $GRAFT_REPO_ROOT; O=gpurun_out/team1; mkdir -p $O
timeout 180 python tools/time_team.py --check --small > $O/check_small.log 2>&1; echo "check_small rc=$?" >> $O/check_small.log
timeout 180 python tools/time_team.py --check > $O/check_big.log 2>&1; echo "check_big rc=$?" >> $O/check_big.log
tail -20 $O/check_small.log $O/check_big.log
if grep -q "rc=124" $O/check_small.log $O/check_big.log; then echo "HANG - stopping"; exit 1; fi
timeout 120 python tools/time_team.py > $O/time_default.log 2>&1
SLAK_TEAM_TRI=0 timeout 120 python tools/time_all.py > $O/time_all_old.log 2>&1
for nb in 2 3 4; do SLAK_TEAM_NB=$nb timeout 120 python tools/time_team.py > $O/time_nb$nb.log 2>&1; done
for d in 1 2 4 3 7; do SLAK_TEAM_DBG=$d timeout 120 python tools/time_team.py > $O/time_dbg$d.log 2>&1; done
cat $O/time_*.log | grep -v Warning
timeout 900 python -m pytest tests/test_fused_launches_gpu.py tests/test_reference_modules_gpu.py tests/test_block_tail_gpu.py tests/test_distributed_gpu.py tests/test_pybind.py -m gpu -q -x --timeout 300 > $O/pytest_new.log 2>&1; tail -15 $O/pytest_new.log
