#!/bin/bash
# Kernel timeline of the headline call (config 2, 8 circuits per vmap call): start / end of every kernel of the last
# two calls, per queue -- where does the time between two join GEMMs go?
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/headline_timeline
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o head -- python3 bench.py --probe-child --steps 3 --warmup 2 "$@" > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gem = [i for i, r in enumerate(rows) if "cgemm_split" in r["Kernel_Name"]]
i0, i1 = gem[-3], gem[-1]
t0 = int(rows[i0]["End_Timestamp"])
with open("$OUT/timeline.txt", "w") as out:
    for r in rows[i0:i1 + 1]:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        line = f"{s:9.1f} {e:9.1f} {e - s:8.1f} us  q{r.get('Queue_Id', '?')} s{r.get('Stream_Id', '?')}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))}  {r['Kernel_Name'][:70]}"
        print(line); out.write(line + "\n")
PY
rm -rf $OUT/kt
