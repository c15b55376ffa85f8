#!/bin/bash
# round 6: new tests (pair folds, ABI collective, vmap over batched inputs, bench regression runs), the default bench with
# the Heisenberg leg, and the driver's command with EIGHT ranks on this one device at default sizes
cd "$(dirname "$0")/../.."
O=gpurun_out/r6d
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" > $O/status.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench default rc=$?" >> $O/status.txt
TCMI_BENCH_OVERSUBSCRIBE=1 timeout 1500 python bench.py --gpus 8 --no-traffic-probe > $O/over8.json 2> $O/over8.err
echo "over8 default rc=$?" >> $O/status.txt
cat $O/status.txt
tail -3 $O/pytest_gpu.log
