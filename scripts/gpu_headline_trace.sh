#!/bin/bash
# ordered kernel list of the timed steps of config 2 (what runs besides the join GEMM): gpu_headline_trace.sh
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/htrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 bench.py --probe-child --steps 3 --warmup 1 ${BATCH:+--batch $BATCH} > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# last GEMM marks the end of the last step; print the kernels between the last two GEMMs
idx = [i for i, r in enumerate(rows) if "cgemm_mfma_kernel<true>" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:110]}")
PY
