#!/bin/bash
# round 5: full GPU test suite + default bench + per-pass VQE profiles
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_full1
rm -rf $OUT; mkdir -p $OUT
timeout 3000 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench.err
tail -c 1500 $OUT/bench.err
python3 - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
v = d["vqe_step"]
print("headline", d["value"], "ms/step", d["ms_per_step"], "roofline frac", d["roofline"]["frac"])
print("vqe ms", v["ms_per_step"], {k: (x.get("frac"), x.get("issue_frac")) for k, x in v["roofline"].items() if isinstance(x, dict) and "frac" in x})
print("spec", v["specialised_kernels"])
print("rqc", d.get("rqc_amplitude", {}).get("contract_s"), "svqa", d.get("sliced_vqa", {}).get("ms_per_value_and_grad"), "mps", d.get("mps_tebd", {}).get("us_per_bond"))
PY
bash scripts/gpu_vqe_profiles.sh r05a > $OUT/prof.log 2>&1
tail -5 $OUT/prof.log
