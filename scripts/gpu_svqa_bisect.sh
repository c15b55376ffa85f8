#!/bin/bash
run() { echo "== $*"; timeout 500 python bench.py $@ --vqe-qubits 0 --rqc-depth 0 --mps-qubits 0 --no-cpu-baseline --no-traffic-probe 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['sliced_vqa']; print(' svqa ms', s['ms_per_value_and_grad'], s['roofline']['host_bound'])"; }
run --no-graph --no-hea-a
run --no-graph
run --no-hea-a
