#!/bin/bash
# round 6: the join GEMM on two f16 pieces (tcmi_cgemm_split_f16) -- error and time next to the three-piece bf16 kernel
mkdir -p gpurun_out/r6r
timeout 600 python scripts/round6/gpu_gemm_f16.py > gpurun_out/r6r/gemm_f16.txt 2>&1
echo "gemm_f16 rc=$?" >> gpurun_out/r6r/status.txt
cat gpurun_out/r6r/gemm_f16.txt
