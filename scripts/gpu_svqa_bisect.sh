#!/bin/bash
run() { echo "== $*"; env "$1" timeout 500 python bench.py ${@:2} --mps-qubits 0 --no-cpu-baseline --no-traffic-probe 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['sliced_vqa']; print(' svqa ms', s['ms_per_value_and_grad'], s['roofline']['host_bound'], 'vqe', (d.get('vqe_step') or {}).get('ms_per_step'), (s.get('one_rank_of_8_sharded') or {}).get('projected_speedup_8_ranks'))"; }
run TCMI_SPECIALIZE=auto
