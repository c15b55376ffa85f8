#!/bin/bash
# round 6: the join kernels' 4 x 4 epilogue on v_pk_fma_f32 lane selects -- tests of the kernels, error / time, stamps
mkdir -p gpurun_out/r6t
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python scripts/round6/gpu_gemm_f16.py > gpurun_out/r6t/gemm_f16.txt 2>&1
grep "B=32\|max|bf16\|Error\|error" gpurun_out/r6t/gemm_f16.txt
F16_PROBE_MODES=113 timeout 900 python scripts/round6/gpu_gemm_f16_modes.py > gpurun_out/r6t/modes.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6t/modes.txt
