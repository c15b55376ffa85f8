#!/bin/bash
# Final captures of a round: gpu_final_profiles.sh <tag>   (run on the GPU box; results under gpurun_out/final_<tag>/)
export TMPDIR=/tmp
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_$TAG
rm -rf $OUT; mkdir -p $OUT
# 1. the default bench line (with its own PMC traffic probe)
python3 bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench.err
# 2. the same command under rocprofv3 kernel trace (program directly after --)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 bench.py --no-traffic-probe > $OUT/${TAG}_bench_default_under_rocprof.json 2> $OUT/kt.err
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_default_kernel_stats.csv
# 3. headline only (timed region of config 2, nothing else): every cgemm launch in the CSV is a batch-8 join
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kh -o head -- python3 bench.py --probe-child --steps 20 --warmup 3 > /dev/null 2> $OUT/kh.err
cp $(find $OUT/kh -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_headline_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
# 4. PMC passes over one VQE step at n=28 d=12 (forward, measurement, cotangent, adjoint kernels)
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $G --output-format csv -d $OUT/pmc$i -o run -- python3 scripts/gpu_vqe_timing.py 28,12,1 > $OUT/pmc_run$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
with open("$OUT/${TAG}_vqe_n28_d12_pmc.txt", "w") as out:
    out.write("# per-dispatch averages (value, dispatches); FETCH_SIZE / WRITE_SIZE in KiB, traffic = (2 FETCH + WRITE) * 1024\n")
    for k in acc:
        if "tcmi" in k:
            line = k + " " + str({c: (round(v / len(cnt[k][c]), 1), len(cnt[k][c])) for c, v in sorted(acc[k].items())})
            out.write(line + "\n")
PY
find $OUT -name "*counter_collection.csv" -delete
rm -rf $OUT/kt $OUT/kh $OUT/pmc[0-9]*
ls -la $OUT
