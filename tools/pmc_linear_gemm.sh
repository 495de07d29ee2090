#!/bin/bash
# PMC counters of slak_linear_gemm (round 5): HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes, FETCH doubled as the guide prescribes for wide streaming reads on
# gfx950) and the SQ instruction mix / waits / LDS conflicts / MFMA busy share -> gpurun_out/sum/pmc_linear_gemm.txt.  --pmc with --kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/sum
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pl_$i; rocprofv3 --pmc $grp --kernel-trace -d /tmp/pl_$i -o p --output-format csv -- python $R/tools/run_linear_gemm.py > /tmp/pl_$i.log 2>&1
done
python - <<'PY' | tee $R/gpurun_out/sum/pmc_linear_gemm.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pl_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "linear_gemm_kernel" not in n: continue
        acc[n.split("(")[0].replace("void ", "").replace("slak::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
def avg(c, k): return sum(c[k]) / max(1, len(c[k])) if c[k] else float("nan")
# algorithmic bytes per launch: <EPI, KS, SPLIT>: K = 16 KS, N = 4 K, M = 128 * (28, 14, 7)^2
rows = {12: 100352, 24: 25088, 48: 6272}
print("# kernel <EPI 1 = GELU / 2 = DGELU, K/16, SPLIT>: HBM bytes = 2 x FETCH_SIZE KiB + WRITE_SIZE KiB; algorithmic = A + B + two [M][N] tensors (GELU: y1 and a out; DGELU: y1 in, dy1 out)")
for k, c in sorted(acc.items()):
    args = [int(v) for v in k[k.index("<") + 1:k.index(">")].split(",")]
    K = 16 * args[1]; N = 4 * K; M = rows.get(args[1], 0)
    alg = M * K * 2 + N * K * 2 + 2 * M * N * 2          # GELU: two outputs; DGELU: y1 in, dy1 out
    rd, wr = 2 * avg(c, "FETCH_SIZE") * 1024, avg(c, "WRITE_SIZE") * 1024
    print("%s: HBM read %.1f MB + write %.1f MB = %.1f MB; algorithmic %.1f MB (x %.2f)" % (k, rd / 1e6, wr / 1e6, (rd + wr) / 1e6, alg / 1e6, (rd + wr) / alg))
    busy = avg(c, "SQ_BUSY_CU_CYCLES"); wave = avg(c, "SQ_WAVE_CYCLES")
    print("    MFMA busy / CU busy cycles %.2f | insts: MFMA %.0f VALU %.0f LDS %.0f SALU %.0f | LDS bank-conflict share %.2f | of wave cycles: waiting on a counter %.2f, issue stalls %.2f, issuing %.2f" % (
        avg(c, "SQ_VALU_MFMA_BUSY_CYCLES") / max(1, busy), avg(c, "SQ_INSTS_MFMA"), avg(c, "SQ_INSTS_VALU"), avg(c, "SQ_INSTS_LDS"), avg(c, "SQ_INSTS_SALU"),
        avg(c, "SQ_LDS_BANK_CONFLICT") / max(1, avg(c, "SQ_LDS_IDX_ACTIVE")), avg(c, "SQ_WAIT_ANY") / max(1, wave), avg(c, "SQ_WAIT_INST_ANY") / max(1, wave), avg(c, "SQ_ACTIVE_INST_ANY") / max(1, wave)))
PY
