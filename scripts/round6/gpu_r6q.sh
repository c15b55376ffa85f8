#!/bin/bash
# round 6: the VQE step with its micro-batches alternating over two streams (again: round 5 measured -1 %)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6q
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
BASE="--no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2 --mps-qubits 0 --rqc-depth 0 --sv-qubits 0 --svqa-qubits 0 --no-hea-a --no-graph --no-heisenberg --vqe-steps 3"
for k in 1 2; do
  timeout 900 python bench.py $BASE --vqe-streams $k > $O/streams$k.json 2> $O/streams$k.err; echo "streams $k rc=$?" >> $O/status.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6q/*.json")):
    d = [json.loads(l) for l in open(f) if l.startswith("{")][0]["vqe_step"]
    print(f, "ms/step %.1f" % d["ms_per_step"], "kernel ms %.1f" % d["roofline"]["step"]["kernel_ms_per_step"], "mem GiB", d["peak_mem_GiB"])
PY
cat $O/status.txt
