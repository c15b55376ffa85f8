#!/bin/bash
# kernel times of the split GEMM (planes pre-pass + MFMA kernel) next to the f32 MFMA GEMM on the join shape
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/gemm_split
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o gs -- python3 scripts/gpu_gemm_split.py > $OUT/run.log 2>&1
cat $OUT/run.log | tail -20
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "tcmi" in r["Name"]:
        print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
find $OUT -name "*kernel_trace.csv" -delete
