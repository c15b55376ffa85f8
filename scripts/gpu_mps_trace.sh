#!/bin/bash
# ordered kernel list of a few TEBD bond updates in the bulk of the chain (config 5): gaps between the launches
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mtrace
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -o t -- python3 bench.py --steps 2 --warmup 1 --vqe-qubits 0 --rqc-depth 0 --mps-chains 0 --mps-sweeps 1 --no-cpu-baseline --no-traffic-probe > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "svd_block_kernel" in r["Kernel_Name"]]
a, b = idx[-34], idx[-31]          # three bulk bonds of the last (timed) sweep
t0 = int(rows[a]["Start_Timestamp"])
prev = None
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  gap {gap:7.1f}  +{(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:90]}")
    prev = e
PY
