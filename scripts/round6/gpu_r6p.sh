#!/bin/bash
# round 6: the shared side-stream pool (tcmi/_streams.py): sliced VQA inside the whole bench and behind two used foreign
# streams; the GPU suite
cd "$(dirname "$0")/../.."
O=gpurun_out/r6p
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
BASE="--no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2 --mps-qubits 0 --rqc-depth 0 --sv-qubits 0 --vqe-qubits 0"
timeout 900 python bench.py $BASE > $O/head_only.json 2> $O/head_only.err; echo "head_only rc=$?" >> $O/status.txt
TCMI_BENCH_BURN_STREAMS=2 timeout 900 python bench.py $BASE --no-hea-a --no-graph > $O/burn2.json 2> $O/burn2.err; echo "burn2 rc=$?" >> $O/status.txt
TCMI_BENCH_BURN_STREAMS=1 timeout 900 python bench.py $BASE --no-hea-a --no-graph > $O/burn1.json 2> $O/burn1.err; echo "burn1 rc=$?" >> $O/status.txt
timeout 900 python bench.py --no-traffic-probe > $O/full.json 2> $O/full.err; echo "full rc=$?" >> $O/status.txt
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/status.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6p/*.json")):
    d = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    print(f, "svqa ms %.2f" % d["sliced_vqa"]["ms_per_value_and_grad"], "headline %.4g" % d["value"], "hea_a", (d.get("hea_a") or {}).get("amplitudes_per_s_per_gpu"), "graph", (d.get("hipgraph_replay") or {}).get("amplitudes_per_s_per_gpu"))
PY
cat $O/status.txt; tail -3 $O/pytest_gpu.log
