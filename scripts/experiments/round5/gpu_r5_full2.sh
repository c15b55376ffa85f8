#!/bin/bash
# round 5, second full capture: default bench + per-pass VQE profiles + headline timeline
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_full2
rm -rf $OUT; mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench.err
python3 - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
v = d["vqe_step"]
print("headline", d["value"], "ms/step", d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
print("vqe ms", v["ms_per_step"], {k: (x.get("frac"), x.get("issue_frac")) for k, x in v["roofline"].items() if isinstance(x, dict) and "frac" in x})
print("spec", v["specialised_kernels"])
s = d.get("sliced_vqa", {})
print("rqc", d.get("rqc_amplitude", {}).get("contract_s"), "svqa", s.get("ms_per_value_and_grad"), s.get("one_rank_of_8_sharded", {}).get("ms_per_value_and_grad"), s.get("one_rank_of_8_sharded", {}).get("projected_speedup_8_ranks"), "mps", d.get("mps_tebd", {}).get("us_per_bond"))
PY
bash scripts/gpu_vqe_profiles.sh r05b > $OUT/prof.log 2>&1
tail -3 $OUT/prof.log
