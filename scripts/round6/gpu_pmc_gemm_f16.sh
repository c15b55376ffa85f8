#!/bin/bash
# PMC passes over the two join kernels (scripts/round6/gemm_f16_driver.py): where the waves' cycles go
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_f16
rm -rf $OUT; mkdir -p $OUT
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $G --output-format csv -d $OUT/p$i -o run -- python3 scripts/round6/gemm_f16_driver.py > $OUT/run$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
with open("$OUT/summary.txt", "w") as out:
    out.write("# per-dispatch averages over 4 launches each (M = N = 4096, K = 128, batch 32, 4 x 4 epilogue)\n")
    for k in acc:
        if "cgemm" in k:
            out.write(k + "\n")
            for c, v in sorted(acc[k].items()):
                out.write("    %-28s %.4g\n" % (c, v / len(cnt[k][c])))
print(open("$OUT/summary.txt").read())
PY
find $OUT -name "*counter_collection.csv" -delete
rm -rf $OUT/p[0-9]*
