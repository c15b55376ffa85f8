#!/bin/bash
# Gaps between consecutive join GEMMs of the headline loop (default bench settings): what the half-circuit chains cost in
# steady state, when the host runs ahead of the device
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/headline_gaps
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o head -- python3 bench.py --probe-child --steps 6 --warmup 2 "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gem = [r for r in rows if "cgemm_split" in r["Kernel_Name"]]
print("launches", len(gem))
for a, b in zip(gem[:-1], gem[1:]):
    gap = (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3
    dur = (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e3
    inside = [r for r in rows if int(a["End_Timestamp"]) <= int(r["Start_Timestamp"]) < int(b["Start_Timestamp"])]
    busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in inside) / 1e3
    print(f"gap {gap:8.1f} us  ({len(inside)} kernels, {busy:7.1f} us of kernel time)   next join {dur:8.1f} us")
PY
rm -rf $OUT/kt
