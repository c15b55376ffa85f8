#!/bin/bash
# round 6: which objective ranks the sliced-VQA trees the way the GPU does?  One seed per run (0..7), measured time next to
# the tree's combo objective and the engine's model time
cd "$(dirname "$0")/../.."
O=gpurun_out/r6i
mkdir -p $O
export TMPDIR=/tmp TCMI_TREE_CACHE=0
ulimit -c 0
COMMON="--sv-qubits 0 --vqe-qubits 0 --mps-qubits 0 --rqc-depth 0 --no-hea-a --no-graph --no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2"
for k in 0 1 2 3 4 5 6 7; do
  timeout 600 python bench.py $COMMON --svqa-seed0 $k > $O/seed_$k.json 2> $O/seed_$k.err; echo "seed $k rc=$?" >> $O/status.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6i/seed_*.json")):
    s = [json.loads(l) for l in open(f) if l.startswith("{")][0]["sliced_vqa"]
    ps = s["path_search"][0]
    print(f, "measured ms %.2f" % s["ms_per_value_and_grad"], "1-of-8 %.2f" % s["one_rank_of_8_sharded"]["ms_per_value_and_grad"],
          "combo %.4g" % ps["objective"][0], "model ms %.2f" % (ps["model_time_s"] * 1e3), {k: round(v, 2) for k, v in s["graphs"].items()},
          "steps", s["steps_per_slice"], s["slice_invariant_steps"], "width", s["contraction_width"])
PY
cat $O/status.txt
