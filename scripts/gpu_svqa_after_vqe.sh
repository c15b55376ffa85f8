#!/bin/bash
# does a preceding VQE leg (with / without plan-specialised kernels) slow the host-bound sliced-VQA leg down?
for spec in 0 auto; do
  echo "== TCMI_SPECIALIZE=$spec"
  TCMI_SPECIALIZE=$spec timeout 500 python bench.py --qubits 16 --vqe-qubits ${1:-28} --vqe-depth ${2:-12} --vqe-batch ${3:-8} --vqe-steps 1 --rqc-depth 0 --mps-qubits 0 --no-cpu-baseline --no-traffic-probe --no-graph --no-hea-a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['sliced_vqa']; print('vqe ms', d['vqe_step']['ms_per_step'], ' svqa ms', s['ms_per_value_and_grad'], s['roofline']['host_bound'])"
done
