#!/bin/bash
# round 6: wider hyper-searches -- sliced VQA with 16 seeds, config 4 with 16 and 32 seeds (one leg per run)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6h
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
COMMON="--sv-qubits 0 --vqe-qubits 0 --mps-qubits 0 --no-hea-a --no-graph --no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2"
timeout 1200 python bench.py $COMMON --rqc-depth 0 --svqa-seeds 16 > $O/svqa_16.json 2> $O/svqa_16.err; echo "svqa 16 rc=$?" >> $O/status.txt
for k in 16 32; do
  timeout 1200 python bench.py $COMMON --svqa-qubits 0 --rqc-seeds $k > $O/rqc_$k.json 2> $O/rqc_$k.err; echo "rqc $k rc=$?" >> $O/status.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6h/*.json")):
    d = [json.loads(l) for l in open(f) if l.startswith("{")][0]
    if "sliced_vqa" in d:
        s = d["sliced_vqa"]
        print(f, "ms %.2f" % s["ms_per_value_and_grad"], "1-of-8 %.2f" % s["one_rank_of_8_sharded"]["ms_per_value_and_grad"], "search", s["path_search_s"], s["graphs"], "value", s["value"])
    if "rqc_amplitude" in d:
        r = d["rqc_amplitude"]
        print(f, "contract ms %.2f" % (r["contract_s"] * 1e3), "search", r["path_search_s"], "model", r["path_search"]["per_seed_model_ms"], "amp", r["amplitude"], "1-of-8", r["time_split"]["one_rank_of_8_sharded_invariants_s"])
PY
cat $O/status.txt
