#!/bin/bash
# round 6: is the slowdown of the sliced-VQA leg behind the headline's extras a collision of streams on hardware queues?
cd "$(dirname "$0")/../.."
O=gpurun_out/r6m
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
BASE="--no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2 --mps-qubits 0 --rqc-depth 0 --sv-qubits 0 --vqe-qubits 0"
run() { tag=$1; shift; env "$@" timeout 900 python bench.py $BASE $EXTRA > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?" >> $O/status.txt; }
EXTRA="" run q4 A=1
EXTRA="" run q8 GPU_MAX_HW_QUEUES=8
EXTRA="--no-hea-a" run q4_nohea A=1
EXTRA="--no-graph" run q4_nograph A=1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6m/*.json")):
    d = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    s = d["sliced_vqa"]
    print(f, "svqa ms %.2f" % s["ms_per_value_and_grad"], "headline %.4g" % d["value"])
PY
cat $O/status.txt
