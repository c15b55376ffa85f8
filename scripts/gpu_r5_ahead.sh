#!/bin/bash
# suffix tables built ahead on a side stream, at 32 circuits per call
mkdir -p gpurun_out/r5_ahead
for a in 0 1 0 1; do
  TCMI_CUT_BUILD_AHEAD=$a timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic-probe --no-hea-a --no-graph \
    --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --svqa-qubits 0 > gpurun_out/r5_ahead/a$a.json 2> gpurun_out/r5_ahead/a$a.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/r5_ahead/a$a.json") if x.startswith("{")]
j=json.loads(l[-1]); print("ahead $a:", j["value"], j["ms_per_step"], j["roofline"]["avg_launch_us"])
PY
done
