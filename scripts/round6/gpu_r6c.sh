#!/bin/bash
# round 6: (1) is the hardware exception what an oversubscribed device does to plain torch?  (2) the whole GPU suite after
# the pruning, (3) the default bench, (4) the driver's command with 4 ranks on this one device at default sizes
cd "$(dirname "$0")/../.."
O=gpurun_out/r6c
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python scripts/round6/gpu_oversub_probe.py > $O/oversub_probe.jsonl 2> $O/oversub_probe.err
echo "oversub probe rc=$?" > $O/status.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" >> $O/status.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench default rc=$?" >> $O/status.txt
TCMI_BENCH_OVERSUBSCRIBE=1 timeout 1200 python bench.py --gpus 4 --no-traffic-probe > $O/over4.json 2> $O/over4.err
echo "over4 default rc=$?" >> $O/status.txt
cat $O/status.txt
tail -3 $O/pytest_gpu.log
