#!/bin/bash
# kernel-time breakdown of any python script: scripts/gpu_stats_any.sh <tag> <script> [args]
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stats_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 ${@:2} > $OUT/run.log 2>&1
tail -2 $OUT/run.log | cut -c1-600
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} total={float(r['TotalDurationNs'])/1e6:9.2f} ms avg={float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}%")
PY
rm -f $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv
