for k in 8 4 3 2; do
echo "MAXK $k"; TCMI_TN_SCAT_MAXK=$k python bench.py --steps 2 --warmup 1 --vqe-qubits 0 --mps-qubits 0 --no-cpu-baseline --no-traffic-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['rqc_amplitude']; print(d['contract_s'], d['tflops'], d['amplitude'], d['roofline'].get('launches'))"
done
