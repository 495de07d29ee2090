#!/bin/bash
# dispatch table, the whole GPU test-suite, the default bench line
cd $GRAFT_REPO_ROOT; O=gpurun_out/suite; mkdir -p $O
timeout 200 python tools/print_dispatch.py > $O/dispatch.json 2> $O/dispatch.err; tail -3 $O/dispatch.err
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; tail -12 $O/pytest.log
timeout 600 python bench.py 2> $O/bench.err | tail -1 > $O/bench.json; cut -c1-1200 $O/bench.json; tail -3 $O/bench.err
