#!/bin/bash
# PMC passes over any python script: scripts/gpu_pmc_script.sh <tag> "<counters pass 1>;<counters pass 2>;..." <script> [args]
# (one rocprofv3 run per ';'-separated counter group; --pmc never combined with trace domains)
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
rm -rf $OUT; mkdir -p $OUT
IFS=';' read -ra GROUPS_ <<< "$2"
i=0
for G in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/p$i -o run -- python3 ${@:3} > $OUT/run_p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
with open("$OUT/summary.txt", "w") as out:
    for k in acc:
        if "tcmi" in k:
            line = k + " " + str({c: (round(v / len(cnt[k][c]), 1), len(cnt[k][c])) for c, v in acc[k].items()})
            print(line); out.write(line + "\n")
PY
find $OUT -name "*counter_collection.csv" -delete
