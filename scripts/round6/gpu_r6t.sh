#!/bin/bash
# round 6: the f16 join kernel with half of a tile's results parked in LDS -- error / time, tests of the kernel, cycle stamps
mkdir -p gpurun_out/r6t
timeout 600 python scripts/round6/gpu_gemm_f16.py > gpurun_out/r6t/gemm_f16.txt 2>&1
echo "gemm_f16 rc=$?" >> gpurun_out/r6t/status.txt
grep "B=32\|max|bf16\|Error\|error" gpurun_out/r6t/gemm_f16.txt
timeout 900 python -m pytest tests/test_gpu_gemm_split.py -x -q -m gpu 2>&1 | tail -3
F16_PROBE_MODES=110 timeout 900 python scripts/round6/gpu_gemm_f16_modes.py > gpurun_out/r6t/modes.txt 2>&1
grep -v amdgpu.ids gpurun_out/r6t/modes.txt
