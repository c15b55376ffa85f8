#!/bin/bash
# round 6: the f16 join in the product path -- its tests, then the headline alone
mkdir -p gpurun_out/r6u
timeout 1500 python -m pytest tests/test_gpu_gemm_split.py -x -q -m gpu > gpurun_out/r6u/pytest_gemm.txt 2>&1
echo "pytest gemm rc=$?" >> gpurun_out/r6u/status.txt
tail -5 gpurun_out/r6u/pytest_gemm.txt
timeout 900 python bench.py --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --svqa-qubits 0 --sv-qubits 0 --no-heisenberg > gpurun_out/r6u/bench_headline.json 2> gpurun_out/r6u/bench_headline.err
echo "bench headline rc=$?" >> gpurun_out/r6u/status.txt
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r6u/bench_headline.json') if x.startswith('{')]
if l:
    j=json.loads(l[-1]); r=j["roofline"]
    print("value %.4g ms/step %.3f frac %.3f kernel %s avg_us %.1f alg %.1f TF hbm_frac %.3f" % (j["value"], j["ms_per_step"], r["frac"], r["kernel"], r["avg_launch_us"], r["algorithmic_achieved"], r.get("hbm_frac_on_algorithmic_bytes", -1)))
    print("join_f32", j.get("join_on_f32_mfma")); print("hea_a", {k: j.get("hea_a", {}).get(k) for k in ("value", "ms_per_step")})
PY
tail -3 gpurun_out/r6u/bench_headline.err
