#!/bin/bash
# round 6: sliced value_and_grad with static leaf buffers / prebuilt slice views in the replay: the leg alone, inside the
# whole default bench, and the tests of the sliced engine
cd "$(dirname "$0")/../.."
O=gpurun_out/r6k
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
COMMON="--sv-qubits 0 --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --no-hea-a --no-graph --no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2"
timeout 600 python bench.py $COMMON > $O/alone.json 2> $O/alone.err; echo "alone rc=$?" >> $O/status.txt
timeout 1500 python -m pytest tests/test_gpu_tn.py tests/test_gpu_multirank.py tests/test_gpu_scale.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/status.txt
timeout 900 python bench.py --no-traffic-probe > $O/full.json 2> $O/full.err; echo "full rc=$?" >> $O/status.txt
python - <<'PY'
import json
for f in ("alone", "full"):
    s = [json.loads(l) for l in open(f"gpurun_out/r6k/{f}.json") if l.startswith("{")][0]["sliced_vqa"]
    print(f, "ms %.2f" % s["ms_per_value_and_grad"], "1-of-8 %.2f" % s["one_rank_of_8_sharded"]["ms_per_value_and_grad"], "cached", s["path_search_cached"], "value", s["value"], s["grad_norm"])
PY
cat $O/status.txt; tail -3 $O/pytest.log
