#!/bin/bash
# round 6: probe modes of the pipelined f16 join kernel (1 no conversion, 2 no MFMA, 5 conversion without plane writes)
mkdir -p gpurun_out/r6t
F16_PROBE_MODES=${F16_PROBE_MODES:-1,2,5,3} F16_PROBE_LIBS=${F16_PROBE_LIBS:-libtcmi_probe.so} timeout 900 python scripts/round6/gpu_gemm_f16_modes.py > gpurun_out/r6t/modes.txt 2>&1
echo "modes rc=$?" >> gpurun_out/r6t/status.txt
grep -v amdgpu.ids gpurun_out/r6t/modes.txt
