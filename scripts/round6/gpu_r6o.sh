#!/bin/bash
# round 6: does the sliced-VQA leg's time with four slice instances depend on WHICH streams of torch's pool it gets?
cd "$(dirname "$0")/../.."
O=gpurun_out/r6o
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
BASE="--no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2 --mps-qubits 0 --rqc-depth 0 --sv-qubits 0 --vqe-qubits 0 --no-hea-a --no-graph"
for k in 0 2 4; do
  TCMI_BENCH_BURN_STREAMS=$k timeout 600 python bench.py $BASE > $O/burn$k.json 2> $O/burn$k.err; echo "burn $k rc=$?" >> $O/status.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6o/*.json")):
    d = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    print(f, "svqa ms %.2f" % d["sliced_vqa"]["ms_per_value_and_grad"])
PY
cat $O/status.txt
