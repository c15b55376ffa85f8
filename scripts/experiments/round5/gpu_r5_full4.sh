#!/bin/bash
# round 5, final capture: full GPU test suite, default bench, rocprofv3 kernel stats of the default bench and of the headline
# alone, per-pass VQE profiles
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5_full4
rm -rf $OUT; mkdir -p $OUT
timeout 2700 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench.err
python3 - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
v = d["vqe_step"]
print("headline", d["value"], "ms/step", d["ms_per_step"], "roofline frac", d["roofline"]["frac"], "traffic", d["roofline"].get("traffic"), "f32 join", d.get("join_on_f32_mfma", {}).get("value"))
print("vqe ms", v["ms_per_step"], {k: (x.get("frac"), x.get("issue_frac")) for k, x in v["roofline"].items() if isinstance(x, dict) and "frac" in x})
print("spec", v["specialised_kernels"], "E", v["mean_energy"], "|g|", v["grad_norm"])
s = d.get("sliced_vqa", {})
print("hea_a", d.get("hea_a", {}).get("amplitudes_per_s_per_gpu"), "rqc", d.get("rqc_amplitude", {}).get("contract_s"), "svqa", s.get("ms_per_value_and_grad"), s.get("one_rank_of_8_sharded", {}).get("ms_per_value_and_grad"), s.get("one_rank_of_8_sharded", {}).get("projected_speedup_8_ranks"), "mps", d.get("mps_tebd", {}).get("us_per_bond"), "cpu", d.get("cpu_baseline", {}).get("value"))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kh -o head -- python3 bench.py --probe-child --steps 20 --warmup 3 > /dev/null 2> $OUT/kh.err
cp $(find $OUT/kh -name "*kernel_stats.csv" | head -1) $OUT/r05d_headline_kernel_stats.csv
rm -rf $OUT/kh
bash scripts/gpu_vqe_profiles.sh r05d > $OUT/prof.log 2>&1
tail -2 $OUT/prof.log
