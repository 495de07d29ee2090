#!/bin/bash
# Every measured artifact of round 6 in ONE GPU session at one commit -> gpurun_out/r06/ (copied into profiles/r06_*)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
git rev-parse HEAD > $O/commit.txt 2>/dev/null
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-mask-bench"
timeout 900 python bench.py                                                          2> $O/bench_cfg1.err | tail -1 > $O/bench_cfg1.json
timeout 600 python bench.py --sparsity 0.4 --no-cpu-baseline --no-mask-bench          2>/dev/null | tail -1 > $O/bench_cfg2_sparsity04.json
timeout 600 python bench.py --model base $Q                                           2>/dev/null | tail -1 > $O/bench_cfg3_slak_b.json
timeout 600 python bench.py --kernel 61 --res 384 $Q                                  2>/dev/null | tail -1 > $O/bench_cfg4_61x61_384px.json
SLAK_WIDE_TRI=0 timeout 600 python bench.py --kernel 61 --res 384 $Q --no-roofline    2>/dev/null | tail -1 > $O/bench_cfg4_wide_tri_off.json
SLAK_TRI_ROWS_SD=0 timeout 600 python bench.py $Q --no-roofline                       2>/dev/null | tail -1 > $O/bench_cfg1_tri_rows_sd_off.json
timeout 600 python bench.py --fp32-dwconv --no-cpu-baseline --no-mask-bench --no-roofline 2>/dev/null | tail -1 > $O/bench_fp32_dwconv_exact.json
# the N > 1 code path on one GPU over RCCL (world size 1), SLaK-T and SLaK-B, with main.py:374-376's own constructor flags beside it
D="$Q --no-roofline --force-dist"
timeout 600 python bench.py $D --sparsity 0.4                                         2>/dev/null | grep '^{' | tail -1 > $O/bench_nccl_ws1.json
timeout 600 python bench.py $D --sparsity 0.4 --ddp-reference-flags                   2>/dev/null | grep '^{' | tail -1 > $O/bench_nccl_ws1_ddp_reference_flags.json
SLAK_GRAD_SLOTS=0 timeout 600 python bench.py $D --sparsity 0.4                       2>/dev/null | grep '^{' | tail -1 > $O/bench_nccl_ws1_no_grad_slots.json
SLAK_BN_BWD_ASYNC=1 timeout 600 python bench.py $D --sparsity 0.4                     2>/dev/null | grep '^{' | tail -1 > $O/bench_nccl_ws1_bn_bwd_async.json
SLAK_FORCE_BN_EXCHANGE=0 timeout 600 python bench.py $D --sparsity 0.4                2>/dev/null | grep '^{' | tail -1 > $O/bench_nccl_ws1_ddp_only.json
timeout 600 python bench.py $D --model base                                           2>/dev/null | grep '^{' | tail -1 > $O/bench_nccl_ws1_slak_b.json
timeout 600 python bench.py $D --model base --ddp-reference-flags                     2>/dev/null | grep '^{' | tail -1 > $O/bench_nccl_ws1_slak_b_ddp_reference_flags.json
SLAK_TUNED_GEMMS=0 OMP_NUM_THREADS=4 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --backend gloo --device 0 --steps 5 --warmup 2 --prime 2 --batch 32 --no-roofline --no-mask-bench --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 > $O/bench_n2_two_ranks_one_gpu_gloo.json
# the step under the profiler: N = 1 and the N > 1 path
BENCH_ARGS="--no-mask-bench" timeout 600 bash tools/profile_bench.sh > /dev/null 2>&1
cp gpurun_out/sum/bench_kernel_stats.txt $O/bench_kernel_stats.txt; cp gpurun_out/sum/step_breakdown.txt $O/step_breakdown.txt; cp gpurun_out/sum/bench_under_rocprof.json $O/bench_under_rocprof.json
python tools/step_gaps.py $(find /tmp/pb -name "*.db" | head -1) --steps 6 --top 25 > $O/step_gaps.txt 2>&1
BENCH_ARGS="--no-mask-bench --no-roofline --force-dist --sparsity 0.4" timeout 600 bash tools/profile_bench.sh > /dev/null 2>&1
cp gpurun_out/sum/step_breakdown.txt $O/step_breakdown_forcedist.txt
python tools/step_gaps.py $(find /tmp/pb -name "*.db" | head -1) --steps 6 --top 25 > $O/step_gaps_forcedist.txt 2>&1
# per-kernel: hardware timestamps, HBM traffic, SQ counters, power
echo "# configs[1]: SLaK-T 51x51, 224 px, 128 images" > $O/kernel_times.txt; timeout 300 bash tools/kernel_times.sh $O/kernel_times.txt
echo "# configs[3]: SLaK-B 51x51, 224 px, 64 images" >> $O/kernel_times.txt; timeout 300 bash tools/kernel_times.sh $O/kernel_times.txt --model base
echo "# configs[4]: SLaK-T 61x61, 384 px, 64 images" >> $O/kernel_times.txt; timeout 300 bash tools/kernel_times.sh $O/kernel_times.txt --kernel 61 --res 384
timeout 900 bash tools/pmc_run.sh > /dev/null 2>&1
cp gpurun_out/sum/pmc_traffic.txt $O/pmc_traffic.txt; cp gpurun_out/sum/pmc_traffic.json $O/pmc_traffic.json
timeout 600 bash tools/pmc_hot.sh > /dev/null 2>&1
cp gpurun_out/sum/pmc_hot.txt $O/pmc_hot.txt
timeout 600 bash tools/pmc_hot.sh --kernel 61 --res 384 > /dev/null 2>&1
cp gpurun_out/sum/pmc_hot.txt $O/pmc_hot_cfg4.txt
timeout 900 bash tools/pmc_step.sh > /dev/null 2>&1
cp gpurun_out/sum/pmc_step.txt $O/pmc_step.txt
timeout 400 python tools/power_trace.py 2>&1 | grep -v amdgpu.ids > $O/power_trace.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json, sys, os
try:
    d = json.load(open(sys.argv[1])); c = d["config"]; r = d.get("roofline") or {}
    print("%-52s %8.1f img/s %7.3f ms  host %6.2f  path_frac %s  frac %s" % (os.path.basename(sys.argv[1]), d["value"], d["ms_per_step"], c.get("host_enqueue_ms_per_step", 0), r.get("path_frac"), r.get("frac")))
except Exception as e:
    print(os.path.basename(sys.argv[1]), "unreadable", e)
PY
done
head -16 $O/step_breakdown.txt | cut -c1-220; tail -3 $O/kernel_times.txt; head -6 $O/step_gaps.txt; head -6 $O/step_gaps_forcedist.txt
