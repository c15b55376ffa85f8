#!/bin/bash
# round 6: the four-product f16 join kernel -- error / time, then MFMAs alone (11), with the fragment reads (12), whole (0)
mkdir -p gpurun_out/r6t
timeout 600 python scripts/round6/gpu_gemm_f16.py > gpurun_out/r6t/gemm_f16.txt 2>&1
echo "gemm_f16 rc=$?" >> gpurun_out/r6t/status.txt
F16_PROBE_MODES=0 timeout 900 python scripts/round6/gpu_gemm_f16_modes.py > gpurun_out/r6t/modes.txt 2>&1
echo "modes rc=$?" >> gpurun_out/r6t/status.txt
grep "B=32\|max|bf16" gpurun_out/r6t/gemm_f16.txt
grep -v amdgpu.ids gpurun_out/r6t/modes.txt
