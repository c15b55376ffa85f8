#!/bin/bash
# round 6: states per vmap call of the n = 28 statevector leg (4 / 8 / 16 of a 16-state step)
mkdir -p gpurun_out/r6ab
for mb in 4 8 16; do
  timeout 600 python bench.py --steps 2 --warmup 1 --sv-microbatch $mb --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --svqa-qubits 0 --no-heisenberg --no-cpu-baseline --no-traffic-probe --no-graph > gpurun_out/r6ab/mb$mb.json 2> gpurun_out/r6ab/mb$mb.err
  python - <<PY
import json
l=[x for x in open('gpurun_out/r6ab/mb$mb.json') if x.startswith('{')]
if l:
    sv=json.loads(l[-1]).get("statevector_n28", {})
    print("microbatch $mb:", sv.get("amplitudes_per_s"), sv.get("ms_per_state"), sv.get("roofline", {}).get("executed_plan", {}).get("frac"), sv.get("skipped"))
else:
    print(open('gpurun_out/r6ab/mb$mb.err').read()[-800:])
PY
done
