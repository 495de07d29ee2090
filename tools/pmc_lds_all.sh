#!/bin/bash
# Dev: LDS bank-conflict share of every hot-path kernel (one PMC pass over tools/time_all.py with few reps).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sum
rm -rf /tmp/pl && SLAK_TIME_ALL_REPS=2 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS --kernel-trace -d /tmp/pl -o p --output-format csv -- python $R/tools/time_all.py > /tmp/pl.log 2>&1
f=$(find /tmp/pl -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY' | tee $R/gpurun_out/sum/pmc_lds_all.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"]
    if "slak::" not in n: continue
    key = n.split("(")[0].replace("void ", "")[:90] + " grid=" + r.get("Grid_Size", "?")
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-110s %12s %12s %8s %10s" % ("kernel", "bank_confl", "lds_active", "ratio", "lds_insts"))
for k, c in sorted(acc.items()):
    bc = sum(c["SQ_LDS_BANK_CONFLICT"]) / max(1, len(c["SQ_LDS_BANK_CONFLICT"]))
    ac = sum(c["SQ_LDS_IDX_ACTIVE"]) / max(1, len(c["SQ_LDS_IDX_ACTIVE"]))
    li = sum(c["SQ_INSTS_LDS"]) / max(1, len(c["SQ_INSTS_LDS"]))
    print("%-110s %12.4g %12.4g %8.3f %10.4g" % (k, bc, ac, bc / ac if ac else 0, li))
PY
