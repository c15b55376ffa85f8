#!/bin/bash
# round 6: the driver's command with four ranks on the one device at default sizes after the weak-scaling default (gloo: launch
# path, sharding, guards -- not a measurement)
mkdir -p gpurun_out/r6z
TCMI_BENCH_OVERSUBSCRIBE=1 timeout 1500 python bench.py --gpus 4 --steps 5 --warmup 2 > gpurun_out/r6z/w4.json 2> gpurun_out/r6z/w4.err
echo "w4 rc=$?" | tee -a gpurun_out/r6z/status.txt
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r6z/w4.json') if x.startswith('{')]
if l:
    j=json.loads(l[-1])
    print({k: j[k] for k in ("value","n_gpus","ms_per_step","scaling")}, j["config"]["global_batch"], j.get("per_rank_ms_per_step"))
    for k in ("statevector_n28","vqe_step","vqe_heisenberg","rqc_amplitude","sliced_vqa","mps_tebd","hea_a"):
        v=j.get(k); print(k, (v.get("skipped") if isinstance(v,dict) and "skipped" in v else "ok") if v is not None else None)
else:
    print(open('gpurun_out/r6z/w4.err').read()[-2000:])
PY
