#!/bin/bash
# round 6: does a hyper-search over seeds give the sliced-VQA leg a better tree?  (1, 4, 8 seeds; only that leg)
cd "$(dirname "$0")/../.."
O=gpurun_out/r6g
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
SV="--sv-qubits 0 --vqe-qubits 0 --rqc-depth 0 --mps-qubits 0 --no-hea-a --no-graph --no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2"
for k in 1 4 8; do
  timeout 900 python bench.py $SV --svqa-seeds $k > $O/svqa_$k.json 2> $O/svqa_$k.err
  echo "svqa seeds $k rc=$?" >> $O/status.txt
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6g/svqa_*.json")):
    d = [json.loads(l) for l in open(f) if l.startswith("{")][0]["sliced_vqa"]
    print(f, "ms %.2f" % d["ms_per_value_and_grad"], "1-of-8 %.2f" % d["one_rank_of_8_sharded"]["ms_per_value_and_grad"],
          "x%.2f" % d["one_rank_of_8_sharded"]["projected_speedup_8_ranks"], "search", d["path_search_s"], d["graphs"],
          d["one_rank_of_8_sharded"]["graphs"], d["one_rank_of_8_sharded"]["invariant_shard_model_us"], "value", d["value"], d["grad_norm"])
PY
cat $O/status.txt
