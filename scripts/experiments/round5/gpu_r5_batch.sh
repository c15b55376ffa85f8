#!/bin/bash
# headline with 8 / 16 / 32 / 64 circuits per call (the step stays 64 circuits)
mkdir -p gpurun_out/r5_batch
for b in 8 16 32 64 8; do
  echo "== batch $b"
  timeout 600 python bench.py --batch $b --steps 10 --warmup 2 --no-cpu-baseline --no-traffic-probe --no-hea-a \
    --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --svqa-qubits 0 > gpurun_out/r5_batch/b$b.json 2> gpurun_out/r5_batch/b$b.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/r5_batch/b$b.json") if x.startswith("{")]
j=json.loads(l[-1]); print(j["value"], j["ms_per_step"], j.get("hipgraph",{}) if isinstance(j.get("hipgraph"),dict) else "")
PY
done
