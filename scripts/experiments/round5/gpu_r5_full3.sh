#!/bin/bash
# round 5, third full capture (after the cotangent fold): default bench, per-pass VQE profiles, the full-size parity prints
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_full3
rm -rf $OUT; mkdir -p $OUT
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench.err
python3 - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
v = d["vqe_step"]
print("headline", d["value"], "ms/step", d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"))
print("vqe ms", v["ms_per_step"], {k: (x.get("frac"), x.get("issue_frac")) for k, x in v["roofline"].items() if isinstance(x, dict) and "frac" in x})
print("dense plan", {k: (x.get("frac") if isinstance(x, dict) else x) for k, x in (v["roofline"].get("dense_plan") or {}).items()})
print("spec", v["specialised_kernels"], "E", v["mean_energy"], "|g|", v["grad_norm"])
s = d.get("sliced_vqa", {})
print("rqc", d.get("rqc_amplitude", {}).get("contract_s"), "svqa", s.get("ms_per_value_and_grad"), s.get("one_rank_of_8_sharded", {}).get("ms_per_value_and_grad"), s.get("one_rank_of_8_sharded", {}).get("projected_speedup_8_ranks"), "mps", d.get("mps_tebd", {}).get("us_per_bond"))
PY
bash scripts/gpu_vqe_profiles.sh r05c > $OUT/prof.log 2>&1
tail -3 $OUT/prof.log
timeout 900 python3 -m pytest tests/test_gpu_scale.py -x -q -m gpu -s -k "config3_full_size" 2>&1 | grep "config 3 full size" > $OUT/config3_parity.txt
cat $OUT/config3_parity.txt
