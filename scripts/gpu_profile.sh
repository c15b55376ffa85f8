#!/bin/bash
# rocprofv3 capture recipe for the bench workload (run on the GPU box from the repo root).
# usage: scripts/gpu_profile.sh <tag> [bench args...]
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT; mkdir -p $OUT
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --rqc-depth 0 ${@:2}"
run() {  # name, rocprof flags...
  name=$1; shift
  rocprofv3 "$@" --output-format csv -d $OUT/$name -o bench -- python3 bench.py $ARGS > $OUT/$name.log 2>&1
  echo "== $name rc=$?"; tail -1 $OUT/$name.log | cut -c1-300
}
run trace --kernel-trace --stats
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run pmc_sq2 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS
run pmc_fetch --pmc FETCH_SIZE
run pmc_write --pmc WRITE_SIZE
find $OUT -type f | head -40
du -sh $OUT
python3 scripts/summarize_prof.py $OUT
