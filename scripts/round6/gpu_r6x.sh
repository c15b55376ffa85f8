#!/bin/bash
# round 6: the headline's weak-scaling default -- two ranks on the one device (gloo, launch path), the multirank bench tests
mkdir -p gpurun_out/r6x
TCMI_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 2 --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --svqa-qubits 0 --sv-qubits 0 --no-heisenberg --no-cpu-baseline --no-traffic-probe > gpurun_out/r6x/w2.json 2> gpurun_out/r6x/w2.err
echo "w2 rc=$?" >> gpurun_out/r6x/status.txt
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r6x/w2.json') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); print({k: j[k] for k in ("value","n_gpus","ms_per_step","scaling")}, j["config"]["global_batch"], j["config"]["workload"][:160], j.get("per_rank_ms_per_step"))
else:
    print(open('gpurun_out/r6x/w2.err').read()[-1500:])
PY
timeout 1500 python -m pytest tests/test_gpu_bench_multirank.py -x -q -m gpu 2>&1 | tail -3
