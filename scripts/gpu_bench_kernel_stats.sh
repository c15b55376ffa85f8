#!/bin/bash
# rocprofv3 --kernel-trace --stats over the default bench command (program directly after --): gpu_bench_kernel_stats.sh <tag>
export TMPDIR=/tmp
TAG=$1
OUT=$GRAFT_REPO_ROOT/gpurun_out/bench_stats_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 bench.py --no-traffic-probe > $OUT/${TAG}_bench_default_under_rocprof.json 2> $OUT/kt.err
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_default_kernel_stats.csv
rm -rf $OUT/kt
head -8 $OUT/${TAG}_bench_default_kernel_stats.csv | cut -c1-160
