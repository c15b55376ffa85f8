#!/bin/bash
# round 6: stream-count theory of the sliced-VQA slowdown behind the headline's extras: instance counts 2 / 3 / 4, and the
# cut contraction on one stream (no side streams created before the leg)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6n
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
BASE="--no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2 --mps-qubits 0 --rqc-depth 0 --sv-qubits 0 --vqe-qubits 0"
run() { tag=$1; shift; env "$@" timeout 900 python bench.py $BASE > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?" >> $O/status.txt; }
run inst2 TCMI_KNOBS=tn_streams_small=2
run inst3 TCMI_KNOBS=tn_streams_small=3
run inst4 A=1
run inst4_cut1 TCMI_KNOBS=cut_streams=0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6n/*.json")):
    d = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    print(f, "svqa ms %.2f" % d["sliced_vqa"]["ms_per_value_and_grad"], "headline %.4g" % d["value"])
PY
cat $O/status.txt
