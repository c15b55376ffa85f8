#!/bin/bash
# round 6: circuits per vmap call of the headline with the f16 join (16 / 32 / 64 / 128 of a 128-circuit step)
mkdir -p gpurun_out/r6v
for b in 16 32 64 128; do
  timeout 600 python bench.py --batch $b --global-batch 128 --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --svqa-qubits 0 --sv-qubits 0 --no-heisenberg --no-cpu-baseline --no-traffic-probe > gpurun_out/r6v/b$b.json 2> gpurun_out/r6v/b$b.err
  python - <<PY
import json
l=[x for x in open('gpurun_out/r6v/b$b.json') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); r=j["roofline"]
    print("batch $b: value %.4g ms/step %.3f join avg_us %.1f per-circuit join us %.1f" % (j["value"], j["ms_per_step"], r["avg_launch_us"], r["avg_launch_us"]/$b))
else:
    print("batch $b: no line"); print(open('gpurun_out/r6v/b$b.err').read()[-600:])
PY
done
