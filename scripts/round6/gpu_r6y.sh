#!/bin/bash
# round 6: the last capture -- GPU suite, the default bench and smoke at the final tree
cd "$(dirname "$0")/../.."
O=gpurun_out/r6y
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
if [ "$1" != "bench-only" ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
  echo "pytest gpu rc=$?" > $O/status.txt
fi
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err
echo "bench default rc=$?" >> $O/status.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
echo "smoke rc=$?" >> $O/status.txt
cat $O/status.txt
grep -E "passed|failed" $O/pytest_gpu.log | tail -1
tail -2 $O/smoke.log
