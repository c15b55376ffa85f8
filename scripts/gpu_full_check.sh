#!/bin/bash
# full GPU test suite + full bench line with a short digest: scripts/gpu_full_check.sh <tag> [bench args]
OUT=gpurun_out/$1; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/full_tests.txt 2>&1; tail -5 $OUT/full_tests.txt
python bench.py --no-traffic-probe ${@:2} > $OUT/bench_full.json 2> $OUT/bench_full_err.txt
python - <<PY
import json
d=json.load(open("$OUT/bench_full.json"))
print("value",d["value"],"ms/step",d["ms_per_step"],"frac",d["roofline"]["frac"],"lat1",d["latency_batch1"]["ms_per_state"], d.get("hipgraph_replay"))
v=d.get("vqe_step")
if v and "error" not in v:
    print("vqe ms/step",v["ms_per_step"],"E",v["mean_energy"],"gn",v["grad_norm"])
    for k,e in v["roofline"].items():
        if e: print(" ",k, {kk:(round(vv,3) if isinstance(vv,float) else vv) for kk,vv in e.items() if kk in ("avg_launch_us","achieved","frac","launches_per_step","kernel_ms_per_step","bound")})
else: print("vqe", v)
r=d.get("rqc_amplitude"); print("rqc", {k:r[k] for k in r if k in ("contract_s","tflops","time_split","error","path_search_s")} if r else None)
m=d.get("mps_tebd"); print("mps",{k:m[k] for k in m if k!="roofline"} if m else None)
print("cpu", d.get("cpu_baseline"))
PY
