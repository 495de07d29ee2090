#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/suite2; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1; tail -8 $O/pytest.log
python bench.py --model base --steps 10 --warmup 3 --no-cpu-baseline --no-mask-bench 2> $O/cfg3.err | tail -1 > $O/bench_cfg3.json; python -c "
import json; d=json.load(open('$O/bench_cfg3.json')); print('cfg3', round(d['value'],1), d['hot_path']['dwconv_frac_of_hbm_peak'], d['hot_path']['lowest_kernel_frac'])"
python bench.py --kernel 61 --res 384 --steps 10 --warmup 3 --no-cpu-baseline --no-mask-bench 2> $O/cfg4.err | tail -1 > $O/bench_cfg4.json; python -c "
import json; d=json.load(open('$O/bench_cfg4.json')); print('cfg4', round(d['value'],1), d['hot_path']['dwconv_frac_of_hbm_peak'], d['hot_path']['lowest_kernel_frac'])"
