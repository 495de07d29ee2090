#!/bin/bash
# The bench lines, the step profile and the stem timings again after the late round-5 changes (stem / downsample side kernels); the hot-path kernels
# are unchanged, so the PMC, power and per-kernel artefacts of tools/round_artifacts.sh stay valid.  -> gpurun_out/r05b/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
timeout 600 python bench.py                                                         2> $O/bench_cfg1.err | tail -1 > $O/bench_cfg1.json
timeout 300 python bench.py --sparsity 0.4 --no-cpu-baseline --no-mask-bench         2>/dev/null | tail -1 > $O/bench_cfg2_sparsity04.json
timeout 300 python bench.py --model base --steps 20 --warmup 5 --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 > $O/bench_cfg3_slak_b.json
timeout 300 python bench.py --kernel 61 --res 384 --steps 20 --warmup 5 --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 > $O/bench_cfg4_61x61_384px.json
BENCH_ARGS="--no-mask-bench" timeout 400 bash tools/profile_bench.sh > /dev/null 2>&1
cp gpurun_out/sum/bench_kernel_stats.txt $O/bench_kernel_stats.txt; cp gpurun_out/sum/step_breakdown.txt $O/step_breakdown.txt; cp gpurun_out/sum/bench_under_rocprof.json $O/bench_under_rocprof.json
timeout 300 python bench.py --force-dist --sparsity 0.4 --steps 20 --warmup 5 --no-roofline --no-mask-bench --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_nccl_ws1.json
timeout 120 python tools/time_stem.py 2>&1 | grep -v amdgpu.ids > $O/stem_times.txt
for f in $O/bench_*.json; do echo "$(basename $f): $(cut -c1-200 $f)"; done
head -16 $O/step_breakdown.txt | cut -c1-200; cat $O/stem_times.txt
