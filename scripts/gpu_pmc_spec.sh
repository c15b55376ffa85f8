#!/bin/bash
# PMC counters of the plan-specialised sweep kernels: scripts/gpu_pmc_spec.sh <tag> <script.py> [script args]
# (one rocprofv3 --pmc run per counter group, kernel-trace/stats only in their own run -- gpurun's rule)
export TMPDIR=/tmp
export TCMI_SPECIALIZE=${TCMI_SPECIALIZE:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_THREAD_CYCLES_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -o p -- python3 $2 ${@:3} > $OUT/run$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
per = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tcmi_spec" in r["Kernel_Name"]:
            per[r["Kernel_Name"][:60]][r["Counter_Name"]][int(r["Dispatch_Id"])] = per[r["Kernel_Name"][:60]][r["Counter_Name"]].get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
with open("$OUT/per_dispatch.txt", "w") as out:
    for k in per:
        for c in sorted(per[k]):
            ids = sorted(per[k][c])
            out.write(k + " " + c + " " + " ".join("%.4g" % per[k][c][i] for i in ids) + "\n")
with open("$OUT/summary.txt", "w") as out:
    for k in acc:
        if "tcmi" in k:
            line = k + " " + str({c: (round(v / len(cnt[k][c]), 1), len(cnt[k][c])) for c, v in sorted(acc[k].items())})
            print(line); out.write(line + "\n")
PY
find $OUT -name "*counter_collection.csv" -delete
