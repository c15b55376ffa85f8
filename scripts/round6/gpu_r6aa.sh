#!/bin/bash
# round 6: what runs between two joins of the headline now that the join takes 1.8 ms (kernel trace of the steady-state loop)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6aa
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o head -- python3 bench.py --probe-child --steps 6 --warmup 2 > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gem = [i for i, r in enumerate(rows) if "cgemm_split" in r["Kernel_Name"]]
a, b = gem[-3], gem[-2]
t0 = int(rows[a]["End_Timestamp"])
print("join", (int(rows[a]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3, "us; between it and the next join:")
for r in rows[a + 1: b + 1]:
    print("  +%7.1f us  %7.1f us  stream %s  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Stream_Id", r.get("Queue_Id", "?")), r["Kernel_Name"][:90]))
PY
rm -rf $OUT/kt
