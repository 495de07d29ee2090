#!/bin/bash
# SQ counters of every hot-path launch (tools/time_all.py, 2 launches each): MFMA busy, LDS bank conflicts, instruction mix.
# Separate rocprofv3 passes per counter group; --pmc with --kernel-trace only.   usage: tools/pmc_hot.sh [time_all args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/sum
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/ph_$i; SLAK_TIME_ALL_REPS=2 rocprofv3 --pmc $grp --kernel-trace -d /tmp/ph_$i -o p --output-format csv -- python $R/tools/time_all.py "$@" > /tmp/ph_$i.log 2>&1
done
python - <<'PY' | tee $R/gpurun_out/sum/pmc_hot.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/ph_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "slak::dwconv" not in n: continue
        key = n.split("(")[0].replace("void ", "").replace("slak::", "")[:84] + " g" + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
def avg(c, k): return sum(c[k]) / max(1, len(c[k]))
print("%-100s %9s %9s %9s %9s %9s %9s" % ("kernel grid", "mfma_util", "lds_confl", "wait_any", "wait_inst", "valu/mfma", "lds/mfma"))
for k, c in sorted(acc.items()):
    busy = avg(c, "SQ_BUSY_CU_CYCLES"); wc = avg(c, "SQ_WAVE_CYCLES")
    print("%-100s %9.3f %9.3f %9.3f %9.3f %9.2f %9.2f" % (k, avg(c, "SQ_VALU_MFMA_BUSY_CYCLES") / (4 * busy) if busy else 0,
          avg(c, "SQ_LDS_BANK_CONFLICT") / max(1, avg(c, "SQ_LDS_IDX_ACTIVE")), avg(c, "SQ_WAIT_ANY") / max(1, wc), avg(c, "SQ_WAIT_INST_ANY") / max(1, wc),
          avg(c, "SQ_INSTS_VALU") / max(1, avg(c, "SQ_INSTS_MFMA")), avg(c, "SQ_INSTS_LDS") / max(1, avg(c, "SQ_INSTS_MFMA"))))
print("mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): share of the CU-busy time its four MFMA pipes are executing; lds_confl = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;")
print("wait_any / wait_inst = share of wave cycles spent waiting for anything / for an instruction issue slot; valu/mfma, lds/mfma = instruction counts per MFMA")
PY
