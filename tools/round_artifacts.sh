#!/bin/bash
# All measured artifacts of a round in one GPU session -> gpurun_out/r02/ (copy what is to be judged into profiles/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
python bench.py                                                         2> $O/bench_cfg1.err | tail -1 > $O/bench_cfg1.json
python bench.py --sparsity 0.4 --no-cpu-baseline --no-mask-bench         2>/dev/null | tail -1 > $O/bench_cfg2_sparsity04.json
python bench.py --sparsity 0.4 --update-frequency 1 --steps 10 --warmup 3 --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 > $O/bench_cfg2_update_every_step.json
python bench.py --model base --steps 20 --warmup 5 --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 > $O/bench_cfg3_slak_b.json
python bench.py --kernel 61 --res 384 --steps 20 --warmup 5 --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 > $O/bench_cfg4_61x61_384px.json
bash tools/profile_bench.sh > /dev/null 2>&1
cp gpurun_out/sum/bench_kernel_stats.txt $O/bench_kernel_stats.txt; cp gpurun_out/sum/step_breakdown.txt $O/step_breakdown.txt; cp gpurun_out/sum/bench_under_rocprof.json $O/bench_under_rocprof.json
bash tools/kernel_times.sh > $O/kernel_times.txt 2>&1
bash tools/pmc_run.sh > /dev/null 2>&1
cp gpurun_out/sum/pmc_traffic.txt $O/pmc_traffic.txt; cp gpurun_out/sum/pmc_traffic.json $O/pmc_traffic.json
for f in $O/bench_cfg*.json; do echo "$(basename $f): $(cut -c1-200 $f)"; done
