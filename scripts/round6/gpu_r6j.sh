#!/bin/bash
# round 6: up to four per-slice graph instances for small slices (sliced value_and_grad): 2 (before) against 4, and the
# tests of the sliced engine
cd "$(dirname "$0")/../.."
O=gpurun_out/r6j
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
COMMON="--sv-qubits 0 --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --no-hea-a --no-graph --no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2"
TCMI_KNOBS=tn_streams_small=2 timeout 600 python bench.py $COMMON > $O/inst2.json 2> $O/inst2.err; echo "inst2 rc=$?" >> $O/status.txt
timeout 600 python bench.py $COMMON > $O/inst4.json 2> $O/inst4.err; echo "inst4 rc=$?" >> $O/status.txt
TCMI_KNOBS=tn_streams_small=8 timeout 600 python bench.py $COMMON > $O/inst8.json 2> $O/inst8.err; echo "inst8 rc=$?" >> $O/status.txt
timeout 1500 python -m pytest tests/test_gpu_tn.py tests/test_gpu_multirank.py tests/test_gpu_bench_multirank.py tests/test_gpu_scale.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/status.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6j/inst*.json")):
    s = [json.loads(l) for l in open(f) if l.startswith("{")][0]["sliced_vqa"]
    print(f, "ms %.2f" % s["ms_per_value_and_grad"], "1-of-8 %.2f" % s["one_rank_of_8_sharded"]["ms_per_value_and_grad"], "value", s["value"], s["grad_norm"])
PY
cat $O/status.txt; tail -3 $O/pytest.log
