#!/bin/bash
# A/B of the VQE leg only (bench.py): scripts/gpu_vqe_ab.sh <tag> VAR=val ...   (runs once with the env as is, once with VAR=val)
OUT=gpurun_out/$1; mkdir -p $OUT
ARGS="--no-traffic-probe --no-cpu-baseline --mps-qubits 0 --rqc-depth 0 --svqa-qubits 0 --no-graph --steps 3 --warmup 1"
python bench.py $ARGS > $OUT/a.json 2> $OUT/a_err.txt
env "${@:2}" python bench.py $ARGS > $OUT/b.json 2> $OUT/b_err.txt
python - <<PY
import json
for k in "ab":
    d=json.load(open("$OUT/%s.json"%k)); v=d["vqe_step"]
    print(k, "vqe ms/step", round(v["ms_per_step"],1), {kk:round(e["avg_launch_us"]) for kk,e in v["roofline"].items() if e and "avg_launch_us" in e})
PY
