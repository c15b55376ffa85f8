#!/bin/bash
# Per-pass captures of the VQE step alone (scripts/gpu_vqe_only.py: only the from-zero value_and_grad of bench.py's
# config-3 leg, one micro-batch of 8): rocprofv3 kernel-trace stats + PMC passes (each counter group in its own run,
# never together with a trace).  The generated kernels are named per pass (tcmi_spec_fwd_p<k>_<digest8> /
# tcmi_spec_adj_p<k>_<digest8>), so every row of both outputs is ONE pass of the executed plan.
#     gpu_vqe_profiles.sh <tag>          -> gpurun_out/vqeprof_<tag>/<tag>_vqe_kernel_stats.csv, <tag>_vqe_pmc.txt
export TMPDIR=/tmp
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/vqeprof_$TAG
rm -rf $OUT; mkdir -p $OUT
N=${2:-28}; D=${3:-12}; B=${4:-8}; STEPS=${5:-3}
python3 scripts/gpu_vqe_only.py $N $D $B $STEPS > $OUT/${TAG}_vqe_events.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o vqe -- python3 scripts/gpu_vqe_only.py $N $D $B $STEPS > $OUT/kt.log 2>&1
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_vqe_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
i=0
for G in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/pmc$i -o run -- python3 scripts/gpu_vqe_only.py $N $D $B 1 > $OUT/pmc_run$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:72]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
with open("$OUT/${TAG}_vqe_pmc.txt", "w") as out:
    out.write("# scripts/gpu_vqe_only.py $N $D $B (two staging calls + 1 timed call of the VQE step, nothing else): per-dispatch\n"
              "# averages (value, dispatches) per kernel = per PASS of the executed plan; FETCH_SIZE / WRITE_SIZE in KiB, HBM traffic per\n"
              "# dispatch = (2 FETCH + WRITE) * 1024 bytes (MI355X_MICROARCH.md: FETCH_SIZE counts 128-byte units on gfx950 in 64-byte\n"
              "# KiB accounting); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles\n")
    for k in sorted(acc):
        if "tcmi" in k:
            out.write(k + " " + str({c: (round(v / len(cnt[k][c]), 1), len(cnt[k][c])) for c, v in sorted(acc[k].items())}) + "\n")
PY
find $OUT -name "*counter_collection.csv" -delete
rm -rf $OUT/kt $OUT/pmc[0-9]*
ls -la $OUT
