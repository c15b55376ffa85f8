#!/bin/bash
# rocprofv3 kernel-trace capture of the default bench command -> profiles/<tag>_kernel_stats.csv + bench line
# usage (on the GPU box): bash scripts/gpu_profile_bench.sh <tag> [bench args]
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 bench.py --no-traffic-probe ${@:2} > $OUT/bench.json 2> $OUT/bench.err
cp $(find $OUT -name "*kernel_stats.csv" | head -1) $OUT/$1_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
head -12 $OUT/$1_kernel_stats.csv
