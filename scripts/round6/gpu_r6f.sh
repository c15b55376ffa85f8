#!/bin/bash
# round 6 captures for profiles/: the GPU suite, the default bench (with the tree cache build() filled), the same command
# under rocprofv3 kernel trace, the VQE step per pass (kernel stats + PMC), the join's PMC counters
cd "$(dirname "$0")/../.."
O=gpurun_out/r6f
mkdir -p $O
export TMPDIR=/tmp
ulimit -c 0
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest gpu rc=$?" > $O/status.txt
timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/bench_default.err
echo "bench default rc=$?" >> $O/status.txt
bash scripts/gpu_bench_kernel_stats.sh r06 > $O/kernel_stats.log 2>&1
echo "kernel stats rc=$?" >> $O/status.txt
bash scripts/gpu_vqe_profiles.sh r06 > $O/vqe_profiles.log 2>&1
echo "vqe profiles rc=$?" >> $O/status.txt
bash scripts/gpu_pmc_split_gemm.sh > $O/pmc_split.log 2>&1
echo "pmc split rc=$?" >> $O/status.txt
cat $O/status.txt
tail -3 $O/pytest_gpu.log
