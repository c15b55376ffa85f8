#!/bin/bash
# round 6: why is the sliced-VQA leg slower inside the whole bench (10.8 ms) than alone (7.7 ms) with four slice instances?
cd "$(dirname "$0")/../.."
O=gpurun_out/r6l
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
BASE="--no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2 --mps-qubits 0 --rqc-depth 0"
run() { tag=$1; shift; timeout 900 python bench.py $BASE "$@" > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?" >> $O/status.txt; }
run vqe_only --sv-qubits 0 --no-hea-a --no-graph --no-heisenberg
run sv_only --vqe-qubits 0 --no-hea-a --no-graph
run heis_only --sv-qubits 0 --no-hea-a --no-graph --vqe-steps 1
run head_only --sv-qubits 0 --vqe-qubits 0
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6l/*.json")):
    s = [json.loads(l) for l in open(f) if l.startswith("{")][0]["sliced_vqa"]
    print(f, "ms %.2f" % s["ms_per_value_and_grad"], "1-of-8 %.2f" % s["one_rank_of_8_sharded"]["ms_per_value_and_grad"])
PY
cat $O/status.txt
