#!/bin/bash
# one PMC pass over a bench configuration: scripts/gpu_pmc_any.sh <tag> "<counters>" [bench args]
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc $2 --output-format csv -d $OUT -o bench -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --svqa-qubits 0 --no-graph --no-traffic-probe ${@:3} > $OUT/run.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:50]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
for k in acc:
    if "tcmi" in k:
        print(k, len(cnt[k]), {c: round(v / len(cnt[k]), 1) for c, v in acc[k].items()})
PY
rm -f $OUT/*/*counter_collection.csv $OUT/*counter_collection.csv
