#!/bin/bash
# round 6, GPU session 7: the whole GPU suite
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -q -m gpu -x --timeout 900 > $O/pytest_all.log 2>&1; tail -15 $O/pytest_all.log | cut -c1-300
