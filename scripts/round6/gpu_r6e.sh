#!/bin/bash
# round 6 experiments on the n = 28 forward passes (statevector leg only): baseline, a pass cap of 48 dense gates (the
# recalibrated pass model's pick), five waves per SIMD, both; and the path search after its loops moved into libtcmi
cd "$(dirname "$0")/../.."
O=gpurun_out/r6e
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
SV="--vqe-qubits 0 --rqc-depth 0 --svqa-qubits 0 --mps-qubits 0 --no-hea-a --no-graph --no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2 --sv-steps 5"
run() { tag=$1; shift; env "$@" TCMI_SPECIALIZE=1 timeout 600 python bench.py $SV > $O/sv_$tag.json 2> $O/sv_$tag.err; echo "sv $tag rc=$?" >> $O/status.txt; }
run base A=1
run cap48 TCMI_KNOBS=pass_cap=48
run cap40 TCMI_KNOBS=pass_cap=40
run waves5 TCMI_KNOBS=spec.waves=5
run waves3 TCMI_KNOBS=spec.waves=3
run base2 A=1
timeout 600 python bench.py --sv-qubits 0 --vqe-qubits 0 --svqa-qubits 0 --mps-qubits 0 --no-hea-a --no-graph --no-traffic-probe --no-cpu-baseline --steps 3 --warmup 2 > $O/rqc.json 2> $O/rqc.err
echo "rqc rc=$?" >> $O/status.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6e/sv_*.json")):
    try:
        d = [json.loads(l) for l in open(f) if l.startswith("{")][0]
        s = d["statevector_n28"]
        print(f, "ms/state %.3f" % s["ms_per_state"], "frac %.3f" % s["roofline"]["executed_plan"]["frac"], "passes", s["passes"],
              "dense frac %.3f" % s["roofline"]["dense_plan"]["frac"], "compiled", s["specialised_kernels"])
    except Exception as e:
        print(f, "failed", e)
d = [json.loads(l) for l in open("gpurun_out/r6e/rqc.json") if l.startswith("{")][0]
print("rqc path_search_s", d["rqc_amplitude"]["path_search_s"], d["rqc_amplitude"]["path_search"]["per_seed_search_s"], "contract_s", d["rqc_amplitude"]["contract_s"])
PY
cat $O/status.txt
