#!/bin/bash
# round 6: the two-piece f16 join with EIGHT waves per workgroup (two per SIMD) -- error / time next to the four-wave kernel,
# cycle stamps
mkdir -p gpurun_out/r6w
timeout 600 python scripts/round6/gpu_gemm_f16.py > gpurun_out/r6w/gemm_f16.txt 2>&1
echo "gemm_f16 rc=$?" >> gpurun_out/r6w/status.txt
grep -v amdgpu.ids gpurun_out/r6w/gemm_f16.txt | grep -v "^   f32\|^   bf16"
F16_PROBE_VARIANT=1 F16_PROBE_MODES=10,110 timeout 900 python scripts/round6/gpu_gemm_f16_modes.py > gpurun_out/r6w/modes.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6w/modes.txt
