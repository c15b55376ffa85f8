#!/bin/bash
# round 6: attribution of the f16 join kernel's time (probe modes, cycle stamps) and its error / time next to the bf16 kernel
mkdir -p gpurun_out/r6s
timeout 600 python scripts/round6/gpu_gemm_f16.py > gpurun_out/r6s/gemm_f16.txt 2>&1
echo "gemm_f16 rc=$?" >> gpurun_out/r6s/status.txt
F16_PROBE_MODES=${F16_PROBE_MODES:-10,110} timeout 900 python scripts/round6/gpu_gemm_f16_modes.py > gpurun_out/r6s/modes.txt 2>&1
echo "modes rc=$?" >> gpurun_out/r6s/status.txt
grep -v amdgpu.ids gpurun_out/r6s/gemm_f16.txt
grep -v amdgpu.ids gpurun_out/r6s/modes.txt
