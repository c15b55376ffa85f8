#!/bin/bash
# deferred crossing gates: kernel + cut tests, then the headline with two / one / no deferred gate
mkdir -p gpurun_out/r5_defer
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -x -q 2>&1 | tail -15
for dfr in 2 1 0; do
  echo "== TCMI_CUT_DEFER=$dfr"
  TCMI_CUT_DEFER=$dfr timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-traffic-probe --no-hea-a \
    --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --svqa-qubits 0 > gpurun_out/r5_defer/d$dfr.json 2> gpurun_out/r5_defer/d$dfr.err
  tail -3 gpurun_out/r5_defer/d$dfr.err
  python - <<PY
import json
l=[x for x in open("gpurun_out/r5_defer/d$dfr.json") if x.startswith("{")]
j=json.loads(l[-1]); print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["avg_launch_us"], j["roofline"]["gemm_shape"], j.get("join_on_f32_mfma"))
PY
done
