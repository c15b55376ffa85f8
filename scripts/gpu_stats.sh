#!/bin/bash
# kernel-time breakdown of a bench.py configuration: scripts/gpu_stats.sh <tag> [bench args]
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/stats_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 ${@:2} > $OUT/run.log 2>&1
tail -1 $OUT/run.log | cut -c1-200
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>5s} total={float(r['TotalDurationNs'])/1e6:8.3f} ms avg={float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}%")
PY
rm -f $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv
