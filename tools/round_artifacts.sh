#!/bin/bash
# All measured artifacts of a round in one GPU session -> gpurun_out/r05/ (copy what is to be judged into profiles/)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05; mkdir -p $O
cd $R
timeout 900 python bench.py                                                         2> $O/bench_cfg1.err | tail -1 > $O/bench_cfg1.json
timeout 600 python bench.py --sparsity 0.4 --no-cpu-baseline --no-mask-bench         2>/dev/null | tail -1 > $O/bench_cfg2_sparsity04.json
timeout 600 python bench.py --model base --steps 20 --warmup 5 --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 > $O/bench_cfg3_slak_b.json
timeout 600 python bench.py --kernel 61 --res 384 --steps 20 --warmup 5 --no-cpu-baseline --no-mask-bench 2>/dev/null | tail -1 > $O/bench_cfg4_61x61_384px.json
timeout 600 python bench.py --fp32-dwconv --no-cpu-baseline --no-mask-bench --no-roofline 2>/dev/null | tail -1 > $O/bench_fp32_dwconv_exact.json
SLAK_TUNED_GEMMS=0 OMP_NUM_THREADS=4 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --device 0 --steps 5 --warmup 2 --prime 2 --batch 32 --no-roofline --no-mask-bench --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_n2_two_ranks_one_gpu_gloo.json
BENCH_ARGS="--no-mask-bench" timeout 600 bash tools/profile_bench.sh > /dev/null 2>&1
cp gpurun_out/sum/bench_kernel_stats.txt $O/bench_kernel_stats.txt; cp gpurun_out/sum/step_breakdown.txt $O/step_breakdown.txt; cp gpurun_out/sum/bench_under_rocprof.json $O/bench_under_rocprof.json
echo "# configs[1]: SLaK-T 51x51, 224 px, 128 images" > $O/kernel_times.txt; timeout 300 bash tools/kernel_times.sh $O/kernel_times.txt
echo "# configs[3]: SLaK-B 51x51, 224 px, 64 images" >> $O/kernel_times.txt; timeout 300 bash tools/kernel_times.sh $O/kernel_times.txt --model base
echo "# configs[4]: SLaK-T 61x61, 384 px, 64 images" >> $O/kernel_times.txt; timeout 300 bash tools/kernel_times.sh $O/kernel_times.txt --kernel 61 --res 384
timeout 900 bash tools/pmc_run.sh > /dev/null 2>&1
cp gpurun_out/sum/pmc_traffic.txt $O/pmc_traffic.txt; cp gpurun_out/sum/pmc_traffic.json $O/pmc_traffic.json
timeout 600 bash tools/pmc_hot.sh > /dev/null 2>&1
cp gpurun_out/sum/pmc_hot.txt $O/pmc_hot.txt
timeout 600 python bench.py --force-dist --sparsity 0.4 --steps 20 --warmup 5 --no-roofline --no-mask-bench --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_nccl_ws1.json
SLAK_LINEAR_GEMM=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-roofline --no-mask-bench --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_cfg1_linear_gemm_off.json
timeout 400 python tools/power_trace.py 2>&1 | grep -v amdgpu.ids > $O/power_trace.txt
timeout 300 python tools/time_gemm2.py 2>&1 | grep -v amdgpu.ids > $O/linear_gemm_times.txt
for f in $O/bench_cfg*.json; do echo "$(basename $f): $(cut -c1-200 $f)"; done
tail -3 $O/kernel_times.txt; head -14 $O/step_breakdown.txt; cat $O/power_trace.txt $O/linear_gemm_times.txt
