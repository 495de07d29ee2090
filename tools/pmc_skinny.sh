#!/bin/bash
# SQ counters of the stage-1 pointwise kernels (tools/time_mlp_fwd.py, tools/time_gelu_bwd.py): instruction mix, waits, LDS conflicts.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/sum
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  for t in time_mlp_fwd time_gelu_bwd; do
    rm -rf /tmp/ps_${i}_$t; rocprofv3 --pmc $grp --kernel-trace -d /tmp/ps_${i}_$t -o p --output-format csv -- python $R/tools/$t.py > /tmp/ps_${i}_$t.log 2>&1
  done
done
python - <<'PY' | tee $R/gpurun_out/sum/pmc_skinny.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/ps_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "slak::linear" not in n and "gelu_bwd_bias" not in n: continue
        acc[n.split("(")[0].replace("void ", "").replace("slak::", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
def avg(c, k): return sum(c[k]) / max(1, len(c[k]))
names = ["SQ_BUSY_CU_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY",
         "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
for k, c in sorted(acc.items()):
    print(k)
    for n in names: print("    %-28s %16.0f" % (n, avg(c, n)))
PY
