#!/bin/bash
# PMC passes over the headline's timed region (join GEMM cgemm_split_kernel): MFMA busy cycles, VALU / LDS instruction
# counts, LDS bank conflicts.  Results: gpurun_out/pmc_split/summary.txt   (one counter group per rocprofv3 run)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_split
rm -rf $OUT; mkdir -p $OUT
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/p$i -o run -- python3 bench.py --probe-child --steps 5 --warmup 2 > $OUT/run$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
with open("$OUT/summary.txt", "w") as out:
    out.write("# per-dispatch averages (value, dispatches) over the headline's timed region (bench.py --probe-child)\n")
    for k in acc:
        if "cgemm" in k or "tcmi_spec_fwd" in k:
            out.write(k + " " + str({c: (round(v / len(cnt[k][c]), 1), len(cnt[k][c])) for c, v in sorted(acc[k].items())}) + "\n")
print(open("$OUT/summary.txt").read())
PY
find $OUT -name "*counter_collection.csv" -delete
rm -rf $OUT/p[0-9]*
